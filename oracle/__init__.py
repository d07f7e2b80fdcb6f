"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the CPU oracle (oracle/vf_oracle.c).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``visfly_amd``) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvf_oracle.so")

ROWS = 28
POS, QUAT, VEL, OMG, MOT, THR, AACC, ACC, T = 0, 3, 7, 10, 13, 17, 21, 24, 27


class Consts(C.Structure):
    """mirror of vfo_consts (oracle/vf_oracle.h)"""
    _fields_ = [
        ("action_type", C.c_int32), ("integrator", C.c_int32),
        ("interval_steps", C.c_int32), ("delay_steps", C.c_int32),
        ("ctrl_delay", C.c_int32), ("trig_mode", C.c_int32),
        ("dt", C.c_float), ("ctrl_dt", C.c_float),
        ("m", C.c_float), ("g_z", C.c_float),
        ("J", C.c_float * 9), ("Jinv", C.c_float * 9),
        ("JP", C.c_float * 9), ("Dm", C.c_float * 9),
        ("B", C.c_float * 16), ("Binv", C.c_float * 16),
        ("c_motor", C.c_float), ("one_minus_c", C.c_float),
        ("tm0", C.c_float), ("tm1", C.c_float), ("tm2", C.c_float),
        ("rot_scale", C.c_float), ("rot_neg_tm1", C.c_float),
        ("rot_tm1sq", C.c_float), ("rot_4tm0", C.c_float),
        ("T_min", C.c_float), ("T_max", C.c_float),
        ("acc_half", C.c_float), ("acc_mean", C.c_float),
        ("rate_half", C.c_float), ("rate_mean", C.c_float),
        ("k_lin", C.c_float * 3), ("k_quad", C.c_float * 3),
        ("wind", C.c_float * 3),
        ("pos_xy_lim", C.c_float), ("pos_z_lo", C.c_float), ("pos_z_hi", C.c_float),
        ("vel_lim", C.c_float), ("omg_lim", C.c_float),
        ("T_init", C.c_float), ("w_init", C.c_float),
        ("vel_half", C.c_float), ("vel_mean", C.c_float),
        ("yaw_half", C.c_float), ("yaw_mean", C.c_float),
        ("vel_p", C.c_float), ("vel_d", C.c_float), ("pos_d", C.c_float),
        ("Pm", C.c_float * 9), ("P12", C.c_float * 9),
    ]


class EnvConsts(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("max_episode_steps", C.c_int32),
        ("is_collision_reset", C.c_int32), ("n_gates", C.c_int32),
        ("bbox_lo", C.c_float * 3), ("bbox_hi", C.c_float * 3),
        ("uav_radius", C.c_float), ("success_radius", C.c_float),
        ("target", C.c_float * 3), ("gates", (C.c_float * 3) * 8),
    ]


class EnvState(C.Structure):
    _fields_ = [
        ("step_count", C.c_void_p), ("reward", C.c_void_p), ("rewards", C.c_void_p),
        ("success", C.c_void_p), ("failure", C.c_void_p), ("episode_done", C.c_void_p),
        ("done", C.c_void_p), ("is_collision", C.c_void_p), ("is_out_bounds", C.c_void_p),
        ("once_collided", C.c_void_p), ("col_point", C.c_void_p), ("col_vec", C.c_void_p),
        ("col_dis", C.c_void_p), ("next_gate", C.c_void_p), ("past_gates", C.c_void_p),
        ("is_pass_next", C.c_void_p),
    ]


# name -> (ctypes field, is_array) ; constants are passed around as a dict of
# float32 numpy scalars/arrays (bit patterns are the parity contract)
GEOMETRIC_FIELDS = ("vel_half", "vel_mean", "yaw_half", "yaw_mean", "vel_p", "vel_d", "pos_d", "Pm", "P12")
CONST_FIELDS = [f[0] for f in Consts._fields_ if f[0] != "trig_mode"]


def consts_from_dict(d):
    c = Consts()
    c.trig_mode = int(d.get("trig_mode", 1))
    for name in CONST_FIELDS:
        if name not in d and name in GEOMETRIC_FIELDS:
            continue      # fixtures of the bodyrate/thrust cases predate the geometric-controller constants
        v = d[name]
        cur = getattr(c, name)
        if isinstance(cur, (int, float)):
            setattr(c, name, v.item() if hasattr(v, "item") else v)
        else:
            arr = np.asarray(v, dtype=np.float32).reshape(-1)
            assert arr.size == len(cur), name
            for i, x in enumerate(arr):
                cur[i] = float(x)
    return c


def build(force=False):
    """compile oracle/libvf_oracle.so with gcc (checker only)"""
    deps = [os.path.join(_HERE, f) for f in ("vf_oracle.c", "vf_oracle.h", "vf_sleef.h")]
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= max(os.path.getmtime(d) for d in deps):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libvf_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        L.vfo_dyn_step.argtypes = [C.POINTER(Consts), C.c_int, fp, fp, ip, fp, fp, fp, fp]
        L.vfo_dyn_step.restype = None
        L.vfo_dyn_reset.argtypes = [C.POINTER(Consts), C.c_int, fp, fp, ip, ip, C.c_int,
                                    fp, fp, fp, fp, fp, fp, fp, fp]
        L.vfo_dyn_reset.restype = None
        L.vfo_update_collision.argtypes = [C.POINTER(EnvConsts), C.c_int, fp,
                                           C.POINTER(EnvState), ip, C.c_int]
        L.vfo_update_collision.restype = None
        L.vfo_env_post_step.argtypes = [C.POINTER(Consts), C.POINTER(EnvConsts), C.c_int, fp,
                                        C.POINTER(EnvState)]
        L.vfo_env_post_step.restype = None
        L.vfo_env_run_steps.argtypes = [C.POINTER(Consts), C.POINTER(EnvConsts), C.c_int, fp, fp, ip, fp, fp, fp, C.c_int,
                                        C.POINTER(EnvState), C.c_int]
        L.vfo_env_run_steps.restype = None
        L.vfo_xmath.argtypes = [C.c_int, fp, fp, fp, C.c_int64]
        L.vfo_xmath.restype = None
        L.vfo_set_threads.argtypes = [C.c_int]
        L.vfo_set_threads.restype = None
        L.vfo_max_threads.argtypes = []
        L.vfo_max_threads.restype = C.c_int
        L.vfo_env_obs.argtypes = [C.POINTER(Consts), C.POINTER(EnvConsts), C.c_int, fp, fp]
        L.vfo_env_obs.restype = None
        L.vfo_env_reset_attr.argtypes = [C.c_int, C.POINTER(EnvState), ip, C.c_int]
        L.vfo_env_reset_attr.restype = None
        L.vfo_gae.argtypes = [fp, fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_double, C.c_double]
        L.vfo_gae.restype = None
        up = C.POINTER(C.c_uint8)
        L.vfo_td_returns.argtypes = [fp, up, up, fp, fp, C.c_int, C.c_int, C.c_double, C.c_double]
        L.vfo_td_returns.restype = None
        _lib = L
    return _lib


def _fp(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class OracleDynamics:
    """Thin numpy wrapper: state slab S [ROWS][N], ring Q [D][4][N]."""

    def __init__(self, consts, N, klin=None, kquad=None):
        self.cd = dict(consts)
        self.c = consts_from_dict(consts)
        self.N = N
        self.S = np.zeros((ROWS, N), np.float32)
        D = int(self.c.delay_steps)
        self.Q = np.zeros((max(D, 1), 4, N), np.float32)
        self.tick = C.c_int32(0)
        self.klin = klin
        self.kquad = kquad
        self.reset()

    def reset(self, pos=None, quat=None, vel=None, omg=None, mot=None, thr=None, t=None,
              idx=None, t_rand=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        pos, quat, vel, omg, mot, thr, t, t_rand = map(f, (pos, quat, vel, omg, mot, thr, t, t_rand))
        if idx is not None:
            idx = np.ascontiguousarray(idx, np.int32)
        lib().vfo_dyn_reset(C.byref(self.c), self.N, _fp(self.S), _fp(self.Q), C.byref(self.tick),
                            _ip(idx), 0 if idx is None else len(idx),
                            _fp(pos), _fp(quat), _fp(vel), _fp(omg), _fp(mot), _fp(thr), _fp(t),
                            _fp(t_rand))

    def set_full_state(self, fs):
        """fs (N,22) = [p q v w motor thrust t] (dynamics.py:793-803)"""
        fs = np.asarray(fs, np.float32)
        self.reset(pos=fs[:, 0:3], quat=fs[:, 3:7], vel=fs[:, 7:10], omg=fs[:, 10:13],
                   mot=fs[:, 13:17], thr=fs[:, 17:21], t=fs[:, 21])

    def step(self, action):
        action = np.ascontiguousarray(action, np.float32)
        obs = np.empty((self.N, 13), np.float32)
        lib().vfo_dyn_step(C.byref(self.c), self.N, _fp(self.S), _fp(self.Q), C.byref(self.tick),
                           _fp(self.klin), _fp(self.kquad), _fp(action), _fp(obs))
        return obs

    @property
    def extend_state(self):
        """(N,28) = [p q v(+wind) w acc ang_acc motor thrust t] (dynamics.py:806-819)"""
        S = self.S
        wind = np.asarray(self.cd["wind"], np.float32).reshape(3, 1)
        return np.concatenate([S[POS:POS + 3], S[QUAT:QUAT + 4], S[VEL:VEL + 3] + wind,
                               S[OMG:OMG + 3], S[ACC:ACC + 3], S[AACC:AACC + 3],
                               S[MOT:MOT + 4], S[THR:THR + 4], S[T:T + 1]], 0).T.copy()


def gae(rewards, values, episode_starts, last_values, dones, gamma, lam):
    T, N = rewards.shape
    adv = np.empty((T, N), np.float32)
    ret = np.empty((T, N), np.float32)
    f = lambda a: np.ascontiguousarray(a, np.float32)
    r, v, e, lv, d = map(f, (rewards, values, episode_starts, last_values, dones))
    lib().vfo_gae(_fp(r), _fp(v), _fp(e), _fp(lv), _fp(d), _fp(adv), _fp(ret), T, N,
                  float(gamma), float(lam))
    return adv, ret


class OracleEnv:
    """CPU restatement of DroneGymEnvsBase.step with visual=False (droneGymEnv.py:141-218):
    dynamics step -> bbox collision -> counters/masks/reward -> (scripted) auto-reset."""

    KINDS = {"hover": 0, "nav": 1, "racing": 2, "hover2": 3, "nav2": 4}

    def __init__(self, consts, N, kind, max_episode_steps, target=(1.0, 0.0, 1.5), success_radius=0.5,
                 is_collision_reset=True, gates=None):
        self.N = N
        self.dyn = OracleDynamics(consts, N)
        e = EnvConsts()
        e.kind = self.KINDS[kind]
        e.max_episode_steps = int(max_episode_steps)
        e.is_collision_reset = int(is_collision_reset)
        lo, hi = (-30.0, -30.0, 0.0), (30.0, 30.0, 8.0)  # droneEnv.py:129
        for d in range(3):
            e.bbox_lo[d], e.bbox_hi[d], e.target[d] = lo[d], hi[d], float(target[d])
        e.uav_radius = 0.1
        e.success_radius = float(success_radius)
        gates = [] if gates is None else gates
        e.n_gates = len(gates)
        for gi, gt in enumerate(gates):
            for d in range(3):
                e.gates[gi][d] = float(gt[d])
        self.e = e
        z = lambda dt, *s: np.zeros(s, dt)
        self.a = dict(step_count=z(np.int32, N), reward=z(np.float32, N), rewards=z(np.float32, N),
                      success=z(np.uint8, N), failure=z(np.uint8, N), episode_done=z(np.uint8, N),
                      done=z(np.uint8, N), is_collision=z(np.uint8, N), is_out_bounds=z(np.uint8, N),
                      once_collided=z(np.uint8, N), col_point=z(np.float32, N, 3), col_vec=z(np.float32, N, 3),
                      col_dis=z(np.float32, N), next_gate=z(np.int32, N), past_gates=z(np.int32, N),
                      is_pass_next=z(np.uint8, N))
        self.es = EnvState()
        for name, _ in EnvState._fields_:
            setattr(self.es, name, self.a[name].ctypes.data)

    def update_collision(self, idx=None):
        if idx is not None:
            idx = np.ascontiguousarray(idx, np.int32)
        lib().vfo_update_collision(C.byref(self.e), self.N, _fp(self.dyn.S), C.byref(self.es), _ip(idx),
                                   0 if idx is None else len(idx))

    def choose_gate(self, idx):
        """RacingEnv._choose_target (envs/RacingEnv.py:172-185)"""
        for i in idx:
            rx = self.dyn.S[POS, i] - np.float32(4.0)
            ry = self.dyn.S[POS + 1, i] - np.float32(0.0)
            self.a["next_gate"][i] = (0 if ry > 0 else 3) if rx < 0 else (1 if rx > 0 else 2)

    def reset_full_state(self, fs):
        """env.reset() with the spawn states given (droneGymEnv.py:302-327; RacingEnv.py:165-170)"""
        self.dyn.set_full_state(fs)
        self.a["once_collided"][:] = 0
        self.update_collision()
        for k in ("reward", "rewards", "done", "episode_done", "step_count"):
            self.a[k][:] = 0
        if self.e.kind == self.KINDS["racing"]:
            self.choose_gate(range(self.N))

    def step(self, action):
        """-> (obs_pre_reset (N,13), reward, done) ; no auto reset here"""
        obs = self.dyn.step(action)
        self.update_collision()
        lib().vfo_env_post_step(C.byref(self.dyn.c), C.byref(self.e), self.N, _fp(self.dyn.S), C.byref(self.es))
        return obs, self.a["reward"].copy(), self.a["done"].copy()

    def run_steps(self, actions, steps):
        """`steps` consecutive step() calls without resets in one OpenMP region (thread-chunked agents): the timing path of
        bench.py's cpu_baseline; arithmetic identical to step().  actions: (n,N,4), used cyclically."""
        actions = np.ascontiguousarray(actions, np.float32).reshape(-1, self.N, 4)
        d = self.dyn
        lib().vfo_env_run_steps(C.byref(d.c), C.byref(self.e), self.N, _fp(d.S), _fp(d.Q), C.byref(d.tick),
                                _fp(d.klin), _fp(d.kquad), _fp(actions), actions.shape[0], C.byref(self.es), int(steps))

    @property
    def obs_state(self):
        """get_observation()["state"] of the env kind (raw state or the HoverEnv2 / NavigationEnv2 variants)"""
        out = np.empty((self.N, 13), np.float32)
        lib().vfo_env_obs(C.byref(self.dyn.c), C.byref(self.e), self.N, _fp(self.dyn.S), _fp(out))
        return out

    def reset_agents(self, idx, fs):
        """reset_agent_by_id with given full states (droneGymEnv.py:339-349, droneEnv.py:260-288)"""
        idx = np.ascontiguousarray(idx, np.int32)
        fs = np.asarray(fs, np.float32)
        if self.e.kind == self.KINDS["racing"]:
            # RacingEnv.reset_agent_by_id (RacingEnv.py:150-163) picks the gate BEFORE the base class
            # re-spawns the agent, i.e. from the terminal position of the finished episode
            self.choose_gate(idx)
            self.a["past_gates"][idx] = 0
            self.a["is_pass_next"][idx] = 0
        self.dyn.reset(pos=fs[:, 0:3], quat=fs[:, 3:7], vel=fs[:, 7:10], omg=fs[:, 10:13], mot=fs[:, 13:17],
                       thr=fs[:, 17:21], t=fs[:, 21], idx=idx)
        self.update_collision(idx)
        self.a["once_collided"][idx] = 0
        lib().vfo_env_reset_attr(self.N, C.byref(self.es), _ip(idx), len(idx))



def xmath(kind, a, b=None):
    """vf_sleef.h over arrays: kind in atan2 | sin | cos | acos (SLEEF u10 routines restated) | sin_cr | cos_cr | acos_cr (fp64
    evaluation rounded once: what the golden generator patches torch.sin / cos / acos to)"""
    k = {"atan2": 0, "sin": 1, "cos": 2, "acos": 3, "sin_cr": 4, "cos_cr": 5, "acos_cr": 6, "atan2_glibc": 7}[kind]
    a = np.ascontiguousarray(a, np.float32)
    b = a if b is None else np.ascontiguousarray(b, np.float32)
    out = np.empty_like(a)
    lib().vfo_xmath(k, _fp(a), _fp(b), _fp(out), a.size)
    return out


def set_threads(n):
    """OpenMP threads used by the following oracle calls (cpu_baseline reports the count it used)"""
    lib().vfo_set_threads(int(n))


def max_threads():
    return int(lib().vfo_max_threads())


def td_returns(r, done, next_value, episode_done=None, gamma=0.99, lamda=0.95):
    H, N = r.shape
    f = lambda a: np.ascontiguousarray(a, np.float32)
    u = lambda a: None if a is None else np.ascontiguousarray(a, np.uint8)
    r, nv, d, ed = f(r), f(next_value), u(done), u(episode_done)
    out = np.empty((H, N), np.float32)
    up = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_uint8))
    lib().vfo_td_returns(_fp(r), up(d), up(ed), _fp(nv), _fp(out), H, N, float(gamma), float(lamda))
    return out
