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
        """e(-G1, sig) * prod_i e(P_i, H(m_i)) for n message hashes (32 bytes each): ONE upload, then on the device
        hash-to-G2 of the hashes, P_i = either the given affine keys (n x 96 bytes) or the per-message key sums
        (key_pts: n x k x 96 bytes, key_scalars: n x k x 32 bytes big-endian) and the (n + 1)-pair multi-pairing;
        576 bytes come back.  The points, hashes and key sums never leave HBM between the three engine calls."""
        import torch
        e = self._eng
        dev = torch.device("cuda", e.device if hasattr(e, "device") else 0)
        parts = [neg_g1, keys_affine if keys_affine is not None else bytes(96 * n), sig, bytes(192 * n), hashes]
        if keys_affine is None:
            parts += [key_pts, key_scalars]
        host = torch.frombuffer(bytearray(b"".join(parts)), dtype=torch.uint8)
        buf = host.to(dev, non_blocking=False)
        o_g1, o_g2 = 0, 96 * (n + 1)
        o_h = o_g2 + 192 * (n + 1)
        o_kp = o_h + 32 * n
        base = buf.data_ptr()
        st = torch.cuda.current_stream(dev).cuda_stream
        if n:
            e._check(e.lib.blsgpu_hash_to_g2_dev(e.h, base + o_h, n, base + o_g2 + 192, st), "blsgpu_hash_to_g2_dev")
            if keys_affine is None:
                inf = torch.empty(n, dtype=torch.uint8, device=dev)
                e._check(e.lib.blsgpu_g1_msm_dev(e.h, base + o_kp, base + o_kp + 96 * n * k, k, n, base + o_g1 + 96, inf.data_ptr(), st),
                         "blsgpu_g1_msm_dev")
        out = torch.empty(576, dtype=torch.uint8, device=dev)
        e.pairing_multi_dev(base + o_g1, base + o_g2, n + 1, out.data_ptr(), st)
        return bytes(out.cpu().numpy())


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
