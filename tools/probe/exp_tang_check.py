"""rj_exp_neg (csrc/hens_rj.h) emulated with exact FMAs against 60-digit arithmetic: constants, table and the maximum error."""
from decimal import Decimal, getcontext
import math, random, struct
getcontext().prec = 60
ln2 = Decimal(2).ln()
L = ln2 / 64
bits = struct.unpack('<Q', struct.pack('<d', float(L)))[0]
hi = struct.unpack('<d', struct.pack('<Q', bits & ~((1 << 21) - 1)))[0]
lo = float(L - Decimal(hi))
INV = float(Decimal(64) / ln2)
tab = [float((ln2 * Decimal(j) / 64).exp()) for j in range(64)]
print("L_hi", hi.hex(), "L_lo", lo.hex(), "INV", INV.hex())
print(",\n".join(", ".join(t.hex() for t in tab[i:i + 4]) for i in range(0, 64, 4)))
fma = lambda a, b, c: float(Decimal(a) * Decimal(b) + Decimal(c))
def myexp(x):
    x = max(x, -800.0)
    kd = float(round(x * INV)); k = int(kd)
    r = fma(-kd, hi, x); r = fma(-kd, lo, r)
    t = fma(r, 1 / 120, 1 / 24); t = fma(r, t, 1 / 6); t = fma(r, t, 0.5)
    q = fma(r * r, t, r)
    return math.ldexp(fma(tab[k & 63], q, tab[k & 63]), k >> 6)
random.seed(1); mx = 0.0
for i in range(200000):
    x = -random.random() * random.choice([1e-3, 0.1, 1, 10, 100, 700])
    ref = Decimal(x).exp()
    if ref > Decimal('1e-300'):
        mx = max(mx, float(abs((Decimal(myexp(x)) - ref) / ref) / Decimal(2.0 ** -52)))
print("max relative error / 2^-52:", mx, "| exp(0) =", myexp(0.0), " exp(-745) =", myexp(-745.0), " exp(-750) =", myexp(-750.0))
