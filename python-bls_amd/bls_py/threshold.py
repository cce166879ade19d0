"""k-of-n threshold helpers (threshold.py:56-136 of the reference)."""
from .bls12381 import n as GROUP_ORDER
from .ec import JacobianPoint, default_ec, generator_Fq
from .fields import Fq
from . import hostmath as H
from .signature import Signature


class Threshold:
    @staticmethod
    def lagrange_coeffs_at_zero(X, ec=default_ec):
        """L_j with P(0) = sum_j L_j P(X[j]) (second barycentric form), as Fq(n, .)."""
        n = ec.n
        k = len(X)
        assert len(set(X)) == k and all(0 != x < n for x in X)
        shifts = []
        for j in range(k):
            w = 1
            for i in range(k):
                if i != j:
                    w = w * (X[j] - X[i]) % n
            shifts.append(pow(w, n - 2, n) * pow(-X[j] % n, n - 2, n) % n)
        den = pow(sum(shifts) % n, n - 2, n)
        return [Fq(n, s * den) for s in shifts]

    @staticmethod
    def interpolate_at_zero(X, Y, ec=default_ec):
        acc = Fq(ec.n, 0)
        for lam, y in zip(Threshold.lagrange_coeffs_at_zero(X, ec), Y):
            acc += lam * y
        return acc

    @staticmethod
    def verify_secret_fragment(T, secret_fragment, player, commitment, ec=default_ec):
        assert len(commitment) == T and secret_fragment != 0 and player != 0
        lhs = generator_Fq(ec) * secret_fragment
        rhs = commitment[0]
        for k in range(1, T):
            rhs = rhs + commitment[k] * pow(player, k, ec.n)
        return lhs == rhs

    @staticmethod
    def aggregate_unit_sigs(signatures, players, T, ec=default_ec):
        """sum_i lambda_i * sig_i  (a |players|-point G2 multi-scalar multiplication)."""
        from .bls import _g2_sum
        lam = Threshold.lagrange_coeffs_at_zero(players, ec)
        return Signature.from_g2(_g2_sum([sig.value for sig in signatures], [int(l) for l in lam]))
