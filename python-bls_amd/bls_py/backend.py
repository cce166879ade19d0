"""Where the host-side scheme code gets its heavy operations from.

Default (and only shipped) provider: the HIP engine behind libblsgpu.so.
`use(provider)` exists so that CPU-only unit tests of the HOST LOGIC can plug in
a stand-in (tests/ inject the CPU oracle); the product never does that and has
no fallback -- with no GPU the default provider raises BlsGpuError.
"""
_provider = None


class HipProvider:
    def __init__(self, device=0):
        from . import _native
        self._eng = _native.engine(device)

    def pairing_multi(self, g1: bytes, g2: bytes, n: int, inf=None) -> bytes:
        return self._eng.pairing_multi(g1, g2, n, inf)

    def miller_loop_batch(self, g1: bytes, g2: bytes, n: int, inf=None) -> bytes:
        return self._eng.miller_loop_batch(g1, g2, n, inf)

    def line_eval_batch(self, r: bytes, q, p: bytes, n: int) -> bytes:
        return self._eng.line_eval_batch(r, q, p, n)

    def final_exp(self, x: bytes) -> bytes:
        return self._eng.final_exp(x)

    def pairing_multi_batch(self, g1: bytes, g2: bytes, gsz: int, groups: int, inf=None) -> bytes:
        return self._eng.pairing_multi_batch(g1, g2, gsz, groups, inf)

    def g1_msm(self, pts: bytes, scalars, k: int, groups: int = 1):
        return self._eng.g1_msm(pts, scalars, k, groups)

    def g2_msm(self, pts: bytes, scalars, k: int, groups: int = 1):
        return self._eng.g2_msm(pts, scalars, k, groups)

    def map_to_g2(self, t: bytes) -> bytes:
        return self._eng.map_to_g2(t)

    def hash_to_g2(self, msg_hashes: bytes) -> bytes:
        return self._eng.hash_to_g2(msg_hashes)

    def g1_decompress(self, data: bytes):
        return self._eng.g1_decompress(data)

    def g2_decompress(self, data: bytes):
        return self._eng.g2_decompress(data)

    # ---- the whole of BLS.verify's device work without a host round trip between its steps (bls.py:153-201) ----
    def verify_pipeline(self, neg_g1: bytes, sig: bytes, hashes: bytes, n: int, keys_affine=None, key_pts=None, key_scalars=None, k=0) -> bytes:
        """e(-G1, sig) * prod_i e(P_i, H(m_i)) for n message hashes (32 bytes each): blsgpu_verify_pipeline -- ONE upload,
        then on the device hash-to-G2 of the hashes, P_i = either the given affine keys (n x 96 bytes) or the per-message
        key sums (key_pts: n x k x 96 bytes, key_scalars: n x k x 32 bytes big-endian) and the (n + 1)-pair
        multi-pairing; 576 bytes come back.  Nothing but the C ABI (no torch)."""
        return self._eng.verify_pipeline(neg_g1, sig, hashes, n, keys_affine, key_pts, key_scalars, k)


def use(provider):
    """Install a provider object with pairing_multi(g1, g2, n, inf=None), final_exp(x),
    miller_loop_batch(g1, g2, n, inf=None), line_eval_batch(r, q|None, p, n),
    g1_msm / g2_msm(pts, scalars|None, k, groups) -> (bytes, [is_inf]),
    map_to_g2(t: n x 192 bytes) -> n x 192 bytes,
    g1_decompress / g2_decompress(bytes) -> (affine bytes, [accepted])."""
    global _provider
    _provider = provider


def get():
    global _provider
    if _provider is None:
        _provider = HipProvider()
    return _provider
