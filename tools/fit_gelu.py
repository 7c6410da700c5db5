"""Fit of the bf16-mode GELU used in the GEMM epilogues (csrc/common.h, gelu_erf_fast):
    gelu(x) ~ x * sigmoid(x * (c0 + c1 x^2 + c2 x^4)),  x^2 clamped at 64
minimax over x in [-9, 9] against 0.5 x (1 + erf(x / sqrt 2)); prints the coefficients in the exp2 domain
(multiplied by -log2 e) and the rounding statistics on the whole bf16 input grid. CPU only (numpy / scipy)."""
import numpy as np
import torch
from scipy.optimize import minimize
from scipy.special import erf

X2MAX = 64.0
L2E = 1.4426950408889634


def gelu(x):
    return 0.5 * x * (1 + erf(x / np.sqrt(2)))


def model(c, x):
    x2 = np.minimum(x * x, X2MAX)
    return x / (1 + np.exp(-((c[2] * x2 + c[1]) * x2 + c[0]) * x))


xs = np.linspace(-9, 9, 36001)
best = (np.inf, np.array([1.595, 7.41e-2, -7.17e-4]))
for _ in range(6):
    r = minimize(lambda c: np.abs(model(c, xs) - gelu(xs)).max(), best[1], method="Nelder-Mead",
                 options=dict(xatol=1e-12, fatol=1e-12, maxiter=4000))
    if r.fun < best[0]:
        best = (r.fun, r.x)
c = best[1]
print("max abs error %.3e" % best[0])
print("exp2-domain coefficients (c0, c1, c2):", ["%.10e" % (-L2E * v) for v in c])

# every bf16 input, f32 arithmetic as in the kernel, result rounded to bf16
allb = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16).float()
xb = allb[(allb.abs() <= 64) & (allb.abs() >= 2 ** -20)]
exact = torch.tensor(gelu(xb.double().numpy())).to(torch.bfloat16)
c32 = [np.float32(-L2E * v) for v in c]
x32 = xb.numpy().astype(np.float32)
x2 = np.minimum(x32 * x32, np.float32(X2MAX))
with np.errstate(over="ignore"):
    e = np.exp2(((c32[2] * x2 + c32[1]) * x2 + c32[0]) * x32).astype(np.float32)
approx = torch.tensor((x32 * (np.float32(1) / (e + np.float32(1)))).astype(np.float32)).to(torch.bfloat16)
big = exact.float().abs() > 1e-3
print("bf16 inputs whose rounded result differs: %d of %d with |gelu| > 1e-3 (max %d ulp)" % (
    ((exact != approx) & big).sum().item(), big.sum().item(),
    (exact.view(torch.int16).int() - approx.view(torch.int16).int()).abs()[big].max().item()))
