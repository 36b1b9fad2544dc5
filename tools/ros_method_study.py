#!/usr/bin/env python3
"""Choice of the Rosenbrock scheme for the stiff device stepper: step counts and outlet accuracy of
candidate schemes on the reference's own DME test case (zNo=20, 0.5 s, golden G4), using the
ORACLE's RHS on the CPU with a dense finite-difference Jacobian ("exact") or the device's
block-bidiagonal one at frozen pressure ("device").  usage: ros_method_study.py [order-check]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs as INP                      # noqa: E402
from oracle import n2_oracle as O         # noqa: E402

KR4 = dict(name="Kaps-Rentrop/Shampine 4(3)", gam=0.5, order=4,
           a=[[], [2.0], [48/25, 6/25], [48/25, 6/25, 0.0]],
           c=[[], [-8.0], [372/25, 12/5], [-112/125, -54/125, -2/5]],
           m=[19/9, 0.5, 25/108, 125/108], e=[17/54, 7/36, 0.0, 125/108], same_f=(3,))
RODAS3 = dict(name="RODAS3 3(2)", gam=0.5, order=3,
              a=[[], [0.0], [2.0, 0.0], [2.0, 0.0, 1.0]],
              c=[[], [4.0], [1.0, -1.0], [1.0, -1.0, -8/3]],
              m=[2.0, 0.0, 1.0, 1.0], e=[0.0, 0.0, 0.0, 1.0], same_f=(1,))
RODAS4 = dict(name="RODAS4 4(3)", gam=0.25, order=4,
              a=[[], [1.544], [0.9466785280815826, 0.2557011698983284],
                 [3.314825187068521, 2.896124015972201, 0.9986419139977817],
                 [1.221224509226641, 6.019134481288629, 12.53708332932087, -0.6878860361058950],
                 [1.221224509226641, 6.019134481288629, 12.53708332932087, -0.6878860361058950, 1.0]],
              c=[[], [-5.6688], [-2.430093356833875, -0.2063599157091915],
                 [-0.1073529058151375, -9.594562251023355, -20.47028614809616],
                 [7.496443313967647, -10.24680431464352, -33.99990352819905, 11.70890893206160],
                 [8.083246795921522, -7.981132988064893, -31.52159432874371, 16.31930543123136, -6.058818238834054]],
              m=[1.221224509226641, 6.019134481288629, 12.53708332932087, -0.6878860361058950, 1.0, 1.0],
              e=[0, 0, 0, 0, 0, 1.0], same_f=())


def step(meth, f, J, y, h):
    n = len(y)
    A = np.linalg.inv(np.eye(n)/(meth["gam"]*h) - J)
    G, fprev = [], None
    for i in range(len(meth["m"])):
        Y = y + sum(a*g for a, g in zip(meth["a"][i], G))
        fi = fprev if i in meth["same_f"] else f(Y)
        fprev = fi
        G.append(A @ (fi + sum(c*g for c, g in zip(meth["c"][i], G))/h))
    return y + sum(m*g for m, g in zip(meth["m"], G)), sum(e*g for e, g in zip(meth["e"], G))


def integrate(meth, f, jac, y0, t0, t1, rtol, atol, h0, hcap):
    t, y, h = t0, y0.copy(), min(h0, hcap)
    nacc = nrej = 0
    p = meth["order"]
    while t < t1:
        last = t + h >= t1
        if last:
            h = t1 - t
        yn, er = step(meth, f, jac(y), y, h)
        worst = np.max(np.abs(er)/(atol + rtol*np.maximum(np.abs(y), np.abs(yn)))) if np.all(np.isfinite(yn)) else 1e300
        if worst <= 1.0:
            t = t1 if last else t + h
            y = yn
            nacc += 1
            fac = min(0.9*max(worst, 1e-10)**(-1.0/p), 5.0)
        else:
            nrej += 1
            fac = max(0.2, 0.9*worst**(-1.0/p)) if worst < 1e299 else 0.25
        h = min(max(h*fac, 1e-14), hcap)
    return y, nacc, nrej


def order_check():
    """van der Pol-like stiff test with known reference by a tiny step: observed order of each scheme"""
    lam = 50.0
    f = lambda y: np.array([y[1], lam*((1 - y[0]**2)*y[1] - y[0])])
    J = lambda y: np.array([[0, 1], [lam*(-2*y[0]*y[1] - 1), lam*(1 - y[0]**2)]])
    y0 = np.array([2.0, 0.0])
    from scipy.integrate import solve_ivp
    ref = solve_ivp(lambda t, y: f(y), (0, 0.5), y0, method="Radau", rtol=1e-13, atol=1e-15, jac=lambda t, y: J(y)).y[:, -1]
    for meth in (KR4, RODAS3, RODAS4):
        errs = []
        for n in (50, 100, 200, 400):
            y, h = y0.copy(), 0.5/n
            for _ in range(n):
                y, _ = step(meth, f, J(y), y, h)
            errs.append(np.max(np.abs(y - ref)))
        print(meth["name"], ["%.2e" % e for e in errs], "orders", ["%.2f" % np.log2(errs[i]/errs[i + 1]) for i in range(3)])


def study():
    name = "dme_script"
    g = np.load(os.path.join(ROOT, "tests", "golden", "g4_tight_%s_lsoda.npz" % name))
    mi = INP.ALL_N2_INPUTS[name]()
    N = 20
    pr = O.setup_n2(mi, N)
    V = pr["varNo"]
    fv = O.make_rhs_vec(pr)
    f = lambda y: fv(0.0, y)

    def jac_dense(y):
        f0 = f(y)
        Jm = np.zeros((V*N, V*N))
        for j in range(V*N):
            d = 1.5e-8*max(abs(y[j]), 1e-3)
            yp = y.copy()
            yp[j] += d
            Jm[:, j] = (f(yp) - f0)/(yp[j] - y[j])
        return Jm

    floc = O.make_local_rhs(pr)
    F1 = pr["vf"]/(pr["BeVoFr"]*pr["zf"])
    FT = pr["vf"]/pr["zf"]

    def jac_device(y):                       # block-bidiagonal, frozen pressure (what the kernel builds)
        Y = y.reshape(1, V, N)
        c, theta, cb, tb, Pn = O.neighbourhood(pr, Y)
        base = floc(c, theta, cb, tb, Pn)[0][:, 0, :]
        st = np.concatenate([c[:, 0, :], theta])
        Jm = np.zeros((V*N, V*N))
        for col in range(V):
            sp = st.copy()
            sp[col] = sp[col] + 1.5e-8*np.maximum(np.abs(st[col]), 1e-3)
            dd = sp[col] - st[col]
            pert = floc(sp[:V - 1].reshape(V - 1, 1, N), sp[V - 1].reshape(1, N), cb, tb, Pn)[0][:, 0, :]
            D = (pert - base)/dd
            for r in range(V):
                for z in range(N):
                    Jm[r*N + z, col*N + z] = D[r, z]
        for i in range(V - 1):
            for z in range(1, N):
                if cb[i, 0, z] > O.EPS_CONST:
                    Jm[i*N + z, i*N + z - 1] += F1*(N - 1)
        for z in range(1, N):
            Jm[(V - 1)*N + z, (V - 1)*N + z - 1] += FT*(N - 1)
        return Jm

    print("| scheme | Jacobian | rtol | accepted | rejected | solves | max rel outlet err (5 output times) |")
    print("|---|---|---|---|---|---|---|")
    for meth in (KR4, RODAS3, RODAS4):
        for jname, jac in (("device", jac_device), ("dense", jac_dense)):
            for rtol in (1e-5, 1e-6, 1e-7, 1e-8):
                y = pr["IV"].copy()
                acc = rej = 0
                errs = []
                h0 = 1e-5
                for k in range(5):
                    t0, t1 = 0.1*k, 0.1*(k + 1)
                    cmax = max(F1, FT)*(N - 1)
                    hcap = min(1.0/(meth["gam"]*cmax), 0.1*(t1 - t0)) if meth is KR4 else 0.25*(t1 - t0)
                    y, a, r = integrate(meth, f, jac, y, t0, t1, rtol, 1e-3*rtol, h0, hcap)
                    acc += a
                    rej += r
                    got = O.pack_interval(y, pr, t1)["dataYs"][:, -1]
                    ref = g["dataYs_%d" % k][:, -1]
                    errs.append(np.max(np.abs(got - ref)/np.abs(ref)))
                print("| %s | %s | %g | %d | %d | %d | %s |" % (meth["name"], jname, rtol, acc, rej,
                      (acc + rej)*len(meth["m"]), " ".join("%.1e" % e for e in errs)), flush=True)


if __name__ == "__main__":
    order_check() if "order-check" in sys.argv else study()
