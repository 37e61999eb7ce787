"""rj_sincos (csrc/hens_rj.h) emulated with exact FMAs against 50-digit arithmetic: two-constant FMA reduction by pi/2 + fdlibm's
kernel polynomials on [-pi/4, pi/4]; maximum absolute error over the arguments the sine leaves produce (|x| <= ~150) and beyond."""
from decimal import Decimal, getcontext
import math, random
getcontext().prec = 50
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494459")
hi = float(PI / 2)
lo = float(PI / 2 - Decimal(hi))
TWO_OVER_PI = float(2 / PI)
print("pi/2 hi", hi.hex(), "lo", lo.hex(), "2/pi", TWO_OVER_PI.hex())
S = [-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04, 2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10]
C = [4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05, -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11]
fma = lambda a, b, c: float(Decimal(a) * Decimal(b) + Decimal(c))
def dsin(x):  # high-precision sin via Taylor after reduction in Decimal
    x = Decimal(x); k = (x / (PI / 2)).to_integral_value(); r = x - k * PI / 2
    def ts(r):
        t, s, n = r, r, 1
        while abs(t) > Decimal("1e-45"): t = -t * r * r / ((2 * n) * (2 * n + 1)); s += t; n += 1
        return s
    def tc(r):
        t, s, n = Decimal(1), Decimal(1), 1
        while abs(t) > Decimal("1e-45"): t = -t * r * r / ((2 * n - 1) * (2 * n)); s += t; n += 1
        return s
    q = int(k) & 3
    s, c = ts(r), tc(r)
    return [(s, c), (c, -s), (-s, -c), (-c, s)][q]
def mysincos(x):
    kd = float(round(x * TWO_OVER_PI)); k = int(kd)
    r = fma(-kd, hi, x); r = fma(-kd, lo, r)
    z = r * r
    t = fma(z, S[5], S[4]); t = fma(z, t, S[3]); t = fma(z, t, S[2]); t = fma(z, t, S[1])
    v = z * r
    s = fma(v, fma(z, t, S[0]), r)
    u = fma(z, C[5], C[4]); u = fma(z, u, C[3]); u = fma(z, u, C[2]); u = fma(z, u, C[1]); u = fma(z, u, C[0])
    hz = 0.5 * z; w = 1.0 - hz
    c = w + (((1.0 - w) - hz) + (z * z) * u)
    return [(s, c), (c, -s), (-s, -c), (-c, s)][k & 3]
random.seed(2)
for span in (1.0, 150.0, 9.0e4):
    ms = mc = 0.0
    for i in range(20000):
        x = (random.random() * 2 - 1) * span
        s, c = mysincos(x); rs, rc = dsin(x)
        ms = max(ms, float(abs(Decimal(s) - rs))); mc = max(mc, float(abs(Decimal(c) - rc)))
    print(f"|x| <= {span:g}: max abs error sin {ms / 2.0**-53:.2f}, cos {mc / 2.0**-53:.2f}  (units of 2^-53)")
