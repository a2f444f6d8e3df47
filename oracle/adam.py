"""Adam, written out (TEST INFRASTRUCTURE -- see oracle/__init__.py).

torch's single-tensor Adam (the implementation the reference's loop runs, homan/jointopt.py:138-151,192) as one fixed
sequence of IEEE fp32 operations per element, with the bias corrections evaluated in double by square-and-multiply:

    m = m + (1 - b1) (g - m);  v = v * b2;  v = v + ((1 - b2) g) g
    step = -(lr / (1 - b1^t));  denom = sqrt(v) / sqrt(1 - b2^t) + eps;  p = p + (step * m) / denom

torch.optim.Adam computes the same quantities through lerp / addcmul / addcdiv kernels whose rounding (fused
multiply-adds, vectorised square roots) differs from host to host; tests/test_objchain.py holds this file within fp32
rounding of torch.optim.Adam.  The HIP step (csrc/adam.hip k_adam) follows the same order, bit for bit.
"""
import numpy as np


def pow_int(b, t):
    """b ** t by square-and-multiply in double: IEEE products in a fixed order (a libm pow may differ in the last bit)."""
    b, r = np.float64(b), np.float64(1.0)
    while t > 0:
        if t & 1:
            r = r * b
        b = b * b
        t >>= 1
    return r


class Adam:
    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8):
        """groups: [{"params": [torch Parameters], "lr": float}] as given to torch.optim.Adam."""
        self.items = [(p, float(g["lr"])) for g in groups for p in g["params"]]
        self.b1, self.b2, self.eps = np.float32(betas[0]), np.float32(betas[1]), np.float32(eps)
        self.state = {}
        self.t = 0

    def zero_grad(self):
        for p, _ in self.items:
            p.grad = None

    def step(self):
        self.t += 1
        f32, f64 = np.float32, np.float64
        bc1 = f64(1.0) - pow_int(f64(self.b1), self.t)
        bc2 = f64(1.0) - pow_int(f64(self.b2), self.t)
        bc2_sqrt = f32(np.sqrt(bc2))
        w1, w2 = f32(f64(1.0) - f64(self.b1)), f32(f64(1.0) - f64(self.b2))
        for p, lr in self.items:
            if p.grad is None:
                continue
            neg_step = f32(-(f64(f32(lr)) / bc1))
            g = p.grad.detach().numpy().astype(f32)
            st = self.state.setdefault(id(p), dict(m=np.zeros_like(g), v=np.zeros_like(g)))
            m, v = st["m"], st["v"]
            m = m + w1 * (g - m)
            v = v * self.b2
            v = v + (w2 * g) * g
            denom = np.sqrt(v) / bc2_sqrt + self.eps
            x = p.detach().numpy()
            x[...] = x + (neg_step * m) / denom
            st["m"], st["v"] = m, v
