"""TEST INFRASTRUCTURE: a stand-in for rmt_app_amd.n2.N2Device backed by the host emulation of the
generated kernel source (oracle/hostemu.py), so that the multi-rank path of rmtExe - partition,
code-object broadcast, per-interval gathers, failure agreement - can run under gloo without a GPU.
Only the fixed-step RK4 entry point is emulated."""
import numpy as np
import torch

from oracle.hostemu import HostEmu
from rmt_app_amd import hipbind, plan

CREATED = []          # (E, code-prefix, sorted defines) of every instance, for the tests to inspect


class EmuDevice:
    def __init__(self, mech, members, N, fp32=False, block=None, npt=None, device=None, extra_opts="",
                 lds_state=None, defines=None, code=None, specialize=None, features=()):
        members = np.ascontiguousarray(members, dtype=np.float64)
        if members.ndim == 1:
            members = members.reshape(1, -1)
        self.mech, self.N, self.E, self.members = mech, int(N), members.shape[0], members
        self.defines = dict(defines or {})
        if specialize is None:
            specialize = self.E >= 2
        if specialize:
            self.defines.update(plan.uniform_member_defines(members, mech.S))
        emu_defs = {k: v for k, v in self.defines.items() if k.startswith("RMT_MC_") or k == "RMT_M2_NEWTON"}
        self.emu = HostEmu(mech.source(hipbind.kernel_template(), defines=emu_defs), tag="emudev", openmp=False)
        self.jacobian_evals = mech.V
        self.flags = np.zeros(self.E, dtype=np.uint32)
        CREATED.append((self.E, bytes(code[:4]) if code else None, tuple(sorted(self.defines))))

    def to_device(self, y):
        return torch.as_tensor(np.ascontiguousarray(y), dtype=torch.float64).reshape(self.E, self.mech.V*self.N).contiguous()

    def rk4(self, y, dt, nsteps, t0=0.0):
        out, fl = self.emu.rk4(y.numpy(), self.members, self.N, dt, nsteps)
        y.copy_(torch.from_numpy(out))
        self.flags |= fl

    def raise_on_flags(self):
        f, self.flags = self.flags, np.zeros(self.E, dtype=np.uint32)
        if f.any():
            raise FloatingPointError("emulated device flags 0x%x in reactor %d" % (int(f.max()), int(np.argmax(f))))

    def rk45_stats(self):
        raise NotImplementedError

    def close(self):
        pass
