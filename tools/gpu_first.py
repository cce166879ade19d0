import sys, json, time
sys.path.insert(0,'python-bls_amd'); sys.path.insert(0,'oracle')
from bls_py import _native
import oracle as O
e=_native.engine(0)
print(e.version())
pj=json.load(open('tests/golden/pairing.json'))
g=pj['gen']
t=time.time(); out=e.pairing_multi(bytes.fromhex(g['g1']), bytes.fromhex(g['g2']), 1); print('gen pairing', time.time()-t, out.hex()==g['final_exp'])
if out.hex()!=g['final_exp']: print(out.hex()[:96]); print(g['final_exp'][:96])
print('final_exp', e.final_exp(bytes.fromhex(g['miller'])).hex()==g['final_exp'])
for k,v in pj['edge'].items():
    n=len(v['g1'])
    out=e.pairing_multi(b''.join(bytes.fromhex(x) for x in v['g1']), b''.join(bytes.fromhex(x) for x in v['g2']), n)
    print(k, out.hex()==v['out'])
v=pj['small4']
out=e.pairing_multi(b''.join(bytes.fromhex(x) for x in v['g1']), b''.join(bytes.fromhex(x) for x in v['g2']), 4)
print('small4', out.hex()==v['out'])
g1=open('tests/golden/pairs_seed1_g1.bin','rb').read(); g2=open('tests/golden/pairs_seed1_g2.bin','rb').read()
for n in (8,65,1025):
    t=time.time(); out=e.pairing_multi(g1[:96*n], g2[:192*n], n); dt=time.time()-t
    print(n, out.hex()==pj['seeded'][str(n)]['out'], 'sec', dt)
for rep in range(3):
    t=time.time(); out=e.pairing_multi(g1, g2, 1025); dt=time.time()-t
    print('1025 again', dt, 1025/dt, 'pairings/s')
