import numpy as np
b1, b2, eps, lr = 0.9, 0.999, 1e-8, 0.002
def alpha(j): return lr*np.sqrt(1-b2**j)/(1-b1**j)
def exact(p, m, v, s, G):
    for k in range(1, G+1):
        m *= b1; v *= b2
        p -= alpha(s+k)*m/(np.sqrt(v)+eps)
    return p, m, v
def approx(p, m, v, s, G, K0):
    p, m, v = exact(p, m, v, s, min(G, K0))
    Gp = G - K0
    if Gp <= 0: return p, m, v
    j1 = s + K0 + 1
    a = np.sqrt(v); g = np.sqrt(b2)
    f1 = alpha(j1)/(a*g+eps); f9 = alpha(j1+8)/(a*g**9+eps)
    rho = (f9/f1)**(1/8)
    q = b1*rho
    R = m*f1*b1*(1-q**Gp)/(1-q)
    return p-R, m*b1**Gp, v*b2**Gp
worst = {}
for K0 in (8, 16, 24, 32):
    w = 0
    for s in (1, 2, 5, 10, 30, 100, 1000, 20000):
        for G in (K0+1, K0+5, K0+20, 200, 5000):
            for gmag in (1e-2, 1e-5, 1e-7, 3e-8, 1e-9):
                m0 = 0.1*gmag; v0 = 0.001*gmag*gmag
                pe, me, ve = exact(0.0, m0, v0, s, G)
                pa, ma, va = approx(0.0, m0, v0, s, G, K0)
                w = max(w, abs(pe-pa))
    worst[K0] = w
print(worst)
# also multi-update history: m/sqrt(v) ~ 1
for K0 in (16, 32):
    w=0
    for s in (1,5,30,1000):
        for G in (K0+1, 100, 3000):
            pe,_,_ = exact(0., 1e-4, 1e-8, s, G); pa,_,_ = approx(0., 1e-4, 1e-8, s, G, K0); w=max(w,abs(pe-pa))
    print(K0, w, pe)

# the rule the kernel uses: more exact steps while lr_t still bends (early steps of training)
def k0_rule(s): return 24 + max(0, 64 - s)
for lr_ in (0.002, 0.02):
    lr = lr_
    w = 0
    for s in (1, 2, 5, 10, 30, 50, 63, 64, 100, 1000, 20000):
        for G in (1, 10, 25, 30, 45, 90, 200, 5000):
            for gmag in (1e-2, 1e-5, 1e-7, 3e-8, 1e-9):
                for ratio in (0.1, 1.0):
                    m0 = ratio * gmag; v0 = (0.001 if ratio == 0.1 else 1.0) * gmag * gmag
                    pe, _, _ = exact(0.0, m0, v0, s, G)
                    pa, _, _ = approx(0.0, m0, v0, s, G, k0_rule(s))
                    w = max(w, abs(pe - pa))
    print("lr", lr_, "worst abs error with the early-step rule", w)
