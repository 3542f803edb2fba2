"""bullet-envs_amd/pybullet_client.py: the reference's INNER seam (Snake(pybullet_client, urdf_root, args), snake.py:14-18).

The reference itself does not travel to the GPU box, so the client is driven here by `SeamLogic`, this file's own
restatement of the calls the reference makes on its client and of what it computes from the answers (each method cites
the lines it follows; in the build container tests/golden/make_env_logic_vectors.py runs the reference's real files
through the same kind of client).  The check: an env-step assembled from the client's single calls -- one
stepSimulation per substep -- is the env-step the fused kernel computes from the same state: the same substep count,
and done flag, and -- with the motor commands rounded to float32 the way the kernel rounds them -- the same observation
bit for bit (both sides run the same substep code: DESIGN.md 3)."""
import numpy as np
import pytest


class SeamLogic(object):
    """What snake.py / SnakeGymEnv.py do with an injected client, in this file's own words."""

    def __init__(self, p, urdf="snake/snake.urdf", f32_targets=False):
        self.p = p
        # the reference multiplies action x pi/6 in float64; the fused kernel does it in float32.  With f32_targets the
        # commands are the kernel's to the last bit, which makes the two sides comparable bit for bit
        self.f32_targets = f32_targets
        p.resetSimulation()                                                  # snake.py:89-93
        p.setAdditionalSearchPath("pybullet_data")
        p.setGravity(0, 0, -9.8)
        p.loadURDF("plane.urdf")
        self.body = p.loadURDF(urdf, [0, 0, 0], useFixedBase=0, flags=p.URDF_USE_SELF_COLLISION)
        friction = [1, 0.1, 0.01]
        p.changeDynamics(self.body, -1, lateralFriction=2, anisotropicFriction=friction)      # snake.py:103-107
        for i in range(p.getNumJoints(self.body)):
            p.changeDynamics(self.body, i, lateralFriction=2, anisotropicFriction=friction)
            p.enableJointForceTorqueSensor(self.body, i, 1)
        self.motors = list(range(3, p.getNumJoints(self.body), 3))          # snake.py:78-81
        self.n = len(self.motors)
        self.scale = np.pi / 6
        self.prev = self.observe()

    def soft_reset(self):                                                    # snake.py:96-99, 119-127
        self.p.resetBasePositionAndOrientation(self.body, [0, 0, 0], [0, 0, 0, 1])
        for j in self.motors:
            self.p.resetJointState(self.body, j, 0)

    def observe(self):                                                       # snake.py:180-217
        n, p = self.n, self.p
        o = np.zeros(3 * n + 8)
        for k, j in enumerate(self.motors):
            o[k] = p.getJointState(self.body, j)[0]
        for k, j in enumerate(self.motors):
            o[n + k] = p.getJointState(self.body, j)[1]
        for k, j in enumerate(self.motors):
            o[2 * n + k] = p.getJointState(self.body, j)[3]
        o[3 * n:3 * n + 3] = p.getBasePositionAndOrientation(self.body)[0]
        o[3 * n + 3:3 * n + 7] = p.getBasePositionAndOrientation(self.body)[1]
        o[3 * n + 7] = p.getJointState(self.body, 0)[2][2]
        return o

    def too_high(self):                                                      # snake.py:237-245
        links = self.p.getLinkStates(self.body, np.arange(0, self.p.getNumJoints(self.body), 3))
        return sum(x[0][2] for x in links) / len(links) > 0.1

    def env_step(self, action, vec_mode):
        a = np.clip(np.asarray(action, dtype=np.float64), -1, 1)            # SnakeGymEnv.py:82-88
        full = np.zeros(self.n)
        full[1::2] = a                                                       # gait 1, snake.py:247-269
        count, high = 0, False
        o = self.observe()
        cmd = full * self.scale
        if self.f32_targets:
            cmd = (full.astype(np.float32) * np.float32(self.scale)).astype(np.float64)
        while np.linalg.norm(cmd - o[:self.n]) > 0.05:                      # snake.py:228-235, 283-304
            self.p.setJointMotorControlArray(self.body, self.motors, self.p.POSITION_CONTROL, list(cmd),
                                             forces=[np.inf] * self.n)
            self.p.stepSimulation()
            o = self.observe()
            count += 1
            if self.too_high():
                high = True
                break
            if count > 40:
                break
        o = self.observe()                                                   # SnakeGymEnv.py:36-42
        n = self.n
        energy = float(np.sum(o[n:2 * n] * o[2 * n:3 * n] * 0.01))
        r = (o[3 * n] - self.prev[3 * n]) + (-10 if abs(o[3 * n + 7]) > 10 else 0) - 0.01 * abs(o[3 * n + 1]) - 0.1 * energy
        done = abs(o[9]) > 0.5 or self.too_high() or high
        if done:
            r += -5
            self.soft_reset()
        self.prev = o
        if done and vec_mode:                                                # multiprocessing_env.py:13-15
            self.soft_reset()
            o = self.prev = self.observe()
        return o, r, done, count


def test_world_building_needs_no_device(pkg):
    """The calls in front of the first state access only collect parameters; what is not on the path raises."""
    p = pkg.BulletClient()
    assert p.connect(p.DIRECT) == 0
    p.resetSimulation()
    p.setGravity(0, 0, -9.8)
    assert p.loadURDF("plane.urdf") == 0
    body = p.loadURDF("snake/snake.urdf", [0, 0, 0], useFixedBase=0, flags=p.URDF_USE_SELF_COLLISION)
    assert p.getNumJoints(body) == 49 and p.getNumJoints(0) == 0
    p.changeDynamics(body, -1, lateralFriction=2, anisotropicFriction=[1, 0.1, 0.01])
    blk = p.loadURDF("../snake/block.urdf", basePosition=[2, 0, 0.1], useFixedBase=0)
    assert blk == 2 and p._world["obstacle"] == 2 and p._world["self_collision"] == 1
    with pytest.raises(NotImplementedError):
        p.loadURDF("r2d2.urdf")
    with pytest.raises(NotImplementedError):
        p.setGravity(1, 0, -9.8)
    with pytest.raises(AttributeError):
        p.applyExternalForce(body, -1, [0, 0, 1], [0, 0, 0], 1)
    with pytest.raises(NotImplementedError):
        p.getLinkStates(body, [1, 2])
    p.disconnect()


class OracleStepper(object):
    """Stands in for _lib.Stepper(1, ...) on a box without a GPU: the calls BulletClient makes on it, answered by the CPU
    oracle (test infrastructure: the product never sees it)."""

    def __init__(self, n_envs, device=0, n_modules=16, **over):
        import oracle as orc
        assert n_envs == 1
        self.e = orc.OracleEnv(n_modules=n_modules, **over)
        self.n = n_modules

    def close(self):
        pass

    def get_state(self):
        tau, fz, px = self.e.get_aux()
        return self.e.get_state()[None, :].astype(np.float64), np.concatenate([tau, [fz, px]])[None, :]

    def set_state(self, s, x=None):
        self.e.set_state(np.asarray(s[0], dtype=np.float64))

    def substep(self, targets, k):
        for _ in range(k):
            self.e.substep(np.asarray(targets[0], dtype=np.float64))

    def link_positions(self):
        return self.e.link_com_world()[1::3][:self.n + 1].T.reshape(1, -1)

    def joint3_reaction_fz(self):
        return np.array([self.e.joint3_reaction_fz()])


def test_client_call_sequence_against_the_oracle(pkg, oracle_mod, monkeypatch):
    """The client's bookkeeping (lazy world, joint / link index maps, target array, caches) with the oracle standing in
    for the device: an env-step assembled from the client's calls is orc_env_step's."""
    import importlib
    mod = importlib.import_module("bullet-envs_amd.pybullet_client")
    monkeypatch.setattr(mod._lib, "Stepper", OracleStepper)
    from test_gpu_env import gait
    p = pkg.BulletClient()
    logic = SeamLogic(p)
    ref = oracle_mod.OracleEnv()
    ref.reset()
    errs = []
    for vec_mode in (True, False):
        for j in range(8):
            a = gait([3], j)[0].astype(np.float64) * (1.2 if j == 4 else 1.0)
            ref.sync(p._stepper().e.get_state(), np.concatenate([p._stepper().e.get_aux()[0], [p._stepper().e.get_aux()[1], logic.prev[48]]]),
                     p._stepper().e.get_manifold())
            o, r, d, k = logic.env_step(a, vec_mode)
            o2, r2, d2, k2, _ = ref.env_step(a, vec_mode=vec_mode)
            assert (k, d) == (k2, d2), (j, k, k2)
            # (the client keeps the motor targets in float32, as the device API takes them: 1e-7 in the outcome)
            kin, dyn = np.r_[0:32, 48:55], np.r_[32:48, 55]          # angles, rates, base pose | impulses / dt (x 240)
            # (one of these env-steps is a stick-slip one that grows the targets' last bit to 5e-2: hence median and cap)
            errs.append(np.abs(o[kin] - o2[kin]).max())
            assert errs[-1] < 0.1 and abs(r - r2) < 1e-2
    assert np.median(errs) < 1e-5, errs
    p.close()


@pytest.mark.gpu
def test_env_step_through_the_client_is_the_fused_kernel(pkg):
    from test_gpu_env import gait
    p = pkg.BulletClient()
    logic = SeamLogic(p, f32_targets=True)
    st = pkg.Stepper(1)
    st.reset()
    same = compared = 0
    for vec_mode in (True, False):
        for j in range(10):
            a = (gait([3], j)[0] * np.float32(1.2 if j == 4 else 1.0)).astype(np.float32)
            # both worlds start the step from the client's state (contact cache included)
            cst = p._stepper()
            S, X = cst.get_state()
            X[0, 17] = np.float32(logic.prev[48])
            st.set_state(S, X)
            st.set_manifold(cst.get_manifold())
            obs, rew, done, sub = st.step(a.reshape(1, 8).copy(), vec_mode=vec_mode)
            o, r, d, k = logic.env_step(a, vec_mode)
            compared += 1
            if k != sub[0]:
                # the servo loop's float32 norm against this file's float64 one: only at the tolerance itself
                assert abs(k - int(sub[0])) == 1
                continue
            assert d == bool(done[0])
            # the same substep code on both sides, the same commands to the last bit: the same observation, bit for bit
            assert np.array_equal(obs[0], o.astype(np.float32)), (j, np.abs(obs[0] - o).max())
            assert abs(float(rew[0]) - r) < 1e-5
            same += 1
    print("client-driven env-steps matching the fused kernel:", same, "of", compared)
    assert same >= compared - 2, (same, compared)
    p.close()
    st.close()
