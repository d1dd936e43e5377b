"""Derive the polynomial used by include/ssx_fmath.h for asin on [0, 0.5].

P(z) ~= (asin(sqrt(z)) - sqrt(z)) / (z*sqrt(z)),  z in [0, 0.25]
Chebyshev fit (mpmath.chebyfit, near-minimax); coefficients printed as C99 hex doubles
and as 17-digit decimals.  Run:  python tools/gen_fmath_coeffs.py [degree]
"""
import sys
import mpmath as mp

mp.mp.dps = 60

def f(z):
    z = mp.mpf(z)
    if z == 0:
        return mp.mpf(1) / 6
    s = mp.sqrt(z)
    return (mp.asin(s) - s) / (z * s)

def main():
    deg = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    coeffs, err = mp.chebyfit(f, [0, mp.mpf(1) / 4], deg + 1, error=True)
    # chebyfit returns highest power first
    coeffs = coeffs[::-1]
    print("// degree", deg, "max abs err of P:", mp.nstr(err, 5))
    for i, c in enumerate(coeffs):
        d = float(c)
        print("  %-24s /* z^%-2d  %.17g */" % (d.hex() + ",", i, d))

if __name__ == "__main__":
    main()
