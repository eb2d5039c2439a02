import numpy as np, sys
sys.path[:0] = ["/root/repo/tests", "/root/repo/tests/golden", "/root/repo"]
import fixture_io
fx = fixture_io.load("c3_e4s2e4")
sd = fx.state_dict() if callable(fx.state_dict) else fx.state_dict
W1 = np.asarray(sd["mlp_predictors.2.0.weight"], np.float64); b1 = np.asarray(sd["mlp_predictors.2.0.bias"], np.float64)
W2 = np.asarray(sd["mlp_predictors.2.2.weight"], np.float64); b2 = np.asarray(sd["mlp_predictors.2.2.bias"], np.float64)
print("W2", W2.shape, "max|W2| %.3g  min nonzero %.3g" % (np.abs(W2).max(), np.abs(W2[W2 != 0]).min()), "b2 max %.3g" % np.abs(b2).max())
rng = np.random.default_rng(0)
x = rng.normal(size=(4096, 7)) * 1.5
h64 = np.tanh(x @ W1.T + b1)
h32 = h64.astype(np.float32)
W32 = W2.astype(np.float32)
exact = h32.astype(np.float64) @ W32.astype(np.float64).T            # exact product of the f32 operands
f32mm = (h32 @ W32.T).astype(np.float64)

def split_f16(v, scale_lo=True):
    hi = v.astype(np.float16)
    r = (v.astype(np.float32) - hi.astype(np.float32))
    lo = (r * np.float32(2048.0 if scale_lo else 1.0)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64) / (2048.0 if scale_lo else 1.0)

def split_bf16_3(v):
    def bf(a):
        u = a.astype(np.float32).view(np.uint32)
        u = ((u.astype(np.uint64) + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
        return u.view(np.float32)
    p0 = bf(v); r = v.astype(np.float32) - p0; p1 = bf(r); p2 = bf(r - p1)
    return p0.astype(np.float64), p1.astype(np.float64), p2.astype(np.float64)

for sl in (True, False):
    hh, hl = split_f16(h32, sl); wh, wl = split_f16(W32, sl)
    approx = hh @ wh.T + hh @ wl.T + hl @ wh.T
    err = np.abs(approx - exact)
    print("f16 x2 (3 products) scale_lo=%s: max abs err %.3g, max rel-to-|w||h| %.3g, rms %.3g" % (sl, err.max(), (err / (np.abs(h32).astype(np.float64) @ np.abs(W32).astype(np.float64).T)).max(), np.sqrt((err**2).mean())))
h0, h1, h2 = split_bf16_3(h32); w0, w1, w2 = split_bf16_3(W32)
approx = h0 @ w0.T + h0 @ w1.T + h1 @ w0.T + h0 @ w2.T + h1 @ w1.T + h2 @ w0.T
err = np.abs(approx - exact)
print("bf16 x3 (6 products): max abs err %.3g, rms %.3g" % (err.max(), np.sqrt((err**2).mean())))
err = np.abs(f32mm - exact)
print("f32 matmul (numpy)  : max abs err %.3g, rms %.3g" % (err.max(), np.sqrt((err**2).mean())))
err = np.abs(h64 @ W2.T - exact)
print("f32 operand rounding vs f64 operands: max abs %.3g rms %.3g" % (err.max(), np.sqrt((err**2).mean())))
print("typical |param| rms %.3g" % np.sqrt((exact**2).mean()))
