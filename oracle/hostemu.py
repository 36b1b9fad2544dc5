"""TEST INFRASTRUCTURE ONLY: build and call the host emulation of the generated kernel source
(see hostemu_driver.cpp).  Used by tests/ and by bench.py's cpu_baseline leg - never by the product.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


class HostEmu:
    def __init__(self, source, tag="emu", openmp=True):
        os.makedirs(_BUILD, exist_ok=True)
        key = hashlib.sha256((source + open(os.path.join(_HERE, "hostemu_driver.cpp")).read()
                              ).encode()).hexdigest()[:16]
        gen = os.path.join(_BUILD, "%s_%s.inc" % (tag, key))
        so = os.path.join(_BUILD, "lib%s_%s%s.so" % (tag, key, "" if openmp else "_st"))
        if not os.path.exists(so):
            uniq = ".%d.tmp" % os.getpid()          # concurrent builders (multi-rank tests) never share a file
            with open(gen + uniq, "w") as f:
                f.write(source)
            os.replace(gen + uniq, gen)
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                   "-DRMT_GENERATED_SOURCE=\"%s\"" % gen,
                   os.path.join(_HERE, "hostemu_driver.cpp"), "-o", so + uniq]
            if openmp:
                cmd.insert(1, "-fopenmp")
            subprocess.run(cmd, check=True, capture_output=True)
            os.replace(so + uniq, so)
        self.lib = C.CDLL(so)
        S, R, V, fp32 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.lib.emu_sizes(C.byref(S), C.byref(R), C.byref(V), C.byref(fp32))
        self.S, self.R, self.V, self.fp32 = S.value, R.value, V.value, bool(fp32.value)
        self.dtype = np.float32 if self.fp32 else np.float64
        vp = C.c_void_p
        self.lib.emu_rhs.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
        self.lib.emu_rhs.restype = None
        self.lib.emu_rk4.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, C.c_longlong, vp]
        self.lib.emu_rk4.restype = None
        self.lib.emu_n1_rhs.argtypes = [vp, vp, vp, C.c_int, vp]
        self.lib.emu_n1_rhs.restype = None
        self.lib.emu_set_threads.argtypes = [C.c_int]
        self.lib.emu_set_threads.restype = C.c_int

    def set_threads(self, n):
        """OpenMP threads used by rhs/rk4 (returns the number in effect; 1 without OpenMP)."""
        return self.lib.emu_set_threads(int(n))

    def rhs(self, y, members, N):
        y = np.ascontiguousarray(y, dtype=self.dtype).reshape(-1, self.V*N)
        E = y.shape[0]
        members = np.ascontiguousarray(members, dtype=np.float64).reshape(E, -1)
        out = np.empty_like(y)
        flags = np.zeros(E, dtype=np.uint32)
        self.lib.emu_rhs(y.ctypes.data, out.ctypes.data, members.ctypes.data, N, E, flags.ctypes.data)
        return out, flags

    def n1_rhs(self, u, members1):
        """modelEquationN1 of the generated source for E states u (E, S+2)."""
        u = np.ascontiguousarray(u, dtype=self.dtype)
        u = u.reshape(-1, u.shape[-1])
        E = u.shape[0]
        members1 = np.ascontiguousarray(members1, dtype=np.float64).reshape(E, -1)
        out = np.empty_like(u)
        flags = np.zeros(E, dtype=np.uint32)
        self.lib.emu_n1_rhs(u.ctypes.data, out.ctypes.data, members1.ctypes.data, E, flags.ctypes.data)
        return out, flags

    def n1_jac(self, u, members1):
        """Model N1 (source generated with defines={"RMT_WITH_N1": "1"}): analytic and forward-difference -d du/d u,
        each [E][V1][V1], and the right-hand sides of rmt_n1_rhs_jac / rmt_n1_rhs, each [E][V1]."""
        u = np.ascontiguousarray(u, dtype=self.dtype)
        E, V1 = u.shape
        members1 = np.ascontiguousarray(members1, dtype=np.float64).reshape(E, -1)
        jan, jfd = np.zeros((E, V1, V1)), np.zeros((E, V1, V1))
        fan, fref = np.zeros((E, V1)), np.zeros((E, V1))
        self.lib.emu_n1_jac.argtypes = [C.c_void_p]*2 + [C.c_int] + [C.c_void_p]*4
        self.lib.emu_n1_jac.restype = None
        self.lib.emu_n1_jac(u.ctypes.data, members1.ctypes.data, E, jan.ctypes.data, jfd.ctypes.data,
                            fan.ctypes.data, fref.ctypes.data)
        return jan, jfd, fan, fref

    def node_jac(self, y, member, N, coupling=False):
        """(analytic, forward-difference) node Jacobians -d f_z/d y_z, each [N][V][V], of one reactor state
        (needs a source generated with defines={"RMT_WITH_ROS4": "1"}); coupling=True (model M2): also the
        (analytic, forward-difference) upwind couplings d f_r/d up_r, each [N][V]."""
        y = np.ascontiguousarray(y, dtype=self.dtype).reshape(self.V*N)
        member = np.ascontiguousarray(member, dtype=np.float64).reshape(-1)
        jan = np.zeros((N, self.V, self.V))
        jfd = np.zeros((N, self.V, self.V))
        lan = np.zeros((N, self.V))
        lfd = np.zeros((N, self.V))
        self.lib.emu_node_jac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]
        self.lib.emu_node_jac.restype = None
        self.lib.emu_node_jac(y.ctypes.data, member.ctypes.data, N, jan.ctypes.data, jfd.ctypes.data,
                              lan.ctypes.data if coupling else None, lfd.ctypes.data if coupling else None)
        return (jan, jfd, lan, lfd) if coupling else (jan, jfd)

    def rk4(self, y, members, N, h, nsteps):
        y = np.array(y, dtype=self.dtype).reshape(-1, self.V*N)
        E = y.shape[0]
        members = np.ascontiguousarray(members, dtype=np.float64).reshape(E, -1)
        flags = np.zeros(E, dtype=np.uint32)
        self.lib.emu_rk4(y.ctypes.data, members.ctypes.data, N, E, float(h), int(nsteps), flags.ctypes.data)
        return y, flags
