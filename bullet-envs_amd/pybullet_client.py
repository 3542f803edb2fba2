"""The INNER seam of the reference: `Snake(pybullet_client, urdf_root, args)` takes its physics engine as an
injected object (/root/reference/snake.py:14-18) and calls ~20 PyBullet functions on it (snake.py:79-330).  This module
is that object, answered by the HIP kernels behind include/snk.h -- one client = one world = one environment, like the
one PyBullet connection each of the reference's env processes owns (SnakeGymEnv.py:23 -> snake.py:89):

    from bullet_envs_amd import BulletClient            # instead of `import pybullet as p`
    p = BulletClient()                                   # p.connect(p.DIRECT) is accepted and does nothing
    robot = Snake(p, "snake/snake.urdf", args)           # the REFERENCE's own snake.py, unmodified
    env = SnakeGymEnv(robot, args)                       # the reference's own SnakeGymEnv.py

Every stepSimulation() is one launch of the single-substep kernel (snk_substep_host) and every read-out a device
synchronisation, so this is a COMPATIBILITY shim -- the reference's Python loop stays the bottleneck (49 getJointState
calls per observation, snake.py:180-206).  The performant seams are the VecEnv (snake_env.py: SnakeVecEnv) and the
device-resident DeviceVecEnv; what this one buys is that the reference's env logic can run on this engine line for
line, which is also how tests check that the fused env-step kernel IS that logic (tests/test_pybullet_client.py).

Only what the path calls is implemented; anything else raises AttributeError (no silent stubs).  The world is created
lazily, at the first call that needs it, from what loadURDF / changeDynamics / setGravity / setTimeStep said before.
"""
import math
import os

import numpy as np

from . import _lib


class BulletClient(object):
    # the constants the path reads off the client
    POSITION_CONTROL = 2
    URDF_USE_SELF_COLLISION = 8
    DIRECT = 2
    GUI = 1

    _PLANE, _SNAKE, _BLOCK = 0, 1, 2

    def __init__(self, device=0, n_modules=16, **params):
        self._device = device
        self._n = n_modules
        self._over = dict(params)
        self._st = None
        self._forget_world()

    # ---- connection / world building (ppo/train.py:60; snake.py:88-107; snake_gait_test.py:44-53) ----
    def connect(self, mode=None, *a, **k):
        return 0

    def disconnect(self, *a, **k):
        self.close()

    def close(self):
        if self._st is not None:
            self._st.close()
            self._st = None

    def _forget_world(self):
        self.close()
        self._world = dict(self._over)       # snk_params overrides collected until the world is built
        self._bodies = {}
        self._targets = None
        self._cache = None

    def resetSimulation(self):
        self._forget_world()

    def setAdditionalSearchPath(self, path):
        return None

    def setGravity(self, x, y, z):
        if x != 0 or y != 0:
            raise NotImplementedError("BulletClient: gravity along z only (snake.py:91)")
        self._set("gravity_z", float(z))

    def setTimeStep(self, dt):                           # snake.py:271-272 (never called on the training path), snake_gait_test.py:53
        self._set("dt", float(dt))

    def _set(self, key, value):
        if self._st is not None and self._world.get(key) != value:
            raise RuntimeError("BulletClient: %s changed after the world was built; call resetSimulation() first" % key)
        self._world[key] = value

    def loadURDF(self, fileName, basePosition=None, baseOrientation=None, useFixedBase=0, flags=0, **kw):
        name = os.path.basename(str(fileName))
        if name == "plane.urdf":
            self._bodies[self._PLANE] = "plane"
            return self._PLANE
        if name.startswith("block"):                     # snake/block.urdf (snake.py:83-84, snake_gait_test.py:51)
            self._set("obstacle", 1 if useFixedBase else 2)
            self._set("obstacle_pos", [float(v) for v in (basePosition or (2.0, 0.0, 0.1))])
            self._bodies[self._BLOCK] = "block"
            return self._BLOCK
        if "snake" in name:
            if basePosition is not None and any(float(v) != 0.0 for v in basePosition):
                raise NotImplementedError("BulletClient: the snake is loaded at the origin (snake.py:93)")
            if useFixedBase:
                raise NotImplementedError("BulletClient: floating base only (snake.py:93)")
            self._set("self_collision", 1 if (flags & self.URDF_USE_SELF_COLLISION) else 0)
            self._bodies[self._SNAKE] = "snake"
            return self._SNAKE
        raise NotImplementedError("BulletClient: no model for %r (plane.urdf, snake.urdf, block.urdf)" % fileName)

    def changeDynamics(self, body, link, lateralFriction=None, anisotropicFriction=None, **kw):
        if kw:
            raise NotImplementedError("BulletClient.changeDynamics: %s" % sorted(kw))
        if body == self._SNAKE:                          # snake.py:103-106: the same values for the base and every link
            if lateralFriction is not None:
                self._set("mu_link", float(lateralFriction))
            if anisotropicFriction is not None:
                self._set("aniso", [float(v) for v in anisotropicFriction])
        elif body == self._BLOCK and lateralFriction is not None:
            self._set("mu_obstacle", float(lateralFriction))

    # what snake_gait_test.py:20-26,54-55,73-74 calls besides: accepted, nothing to do on this engine (no GUI, no real-time
    # clock); getCameraImage hands back PyBullet's 5-tuple with no pixels
    def setRealTimeSimulation(self, enable):
        if enable:
            raise NotImplementedError("BulletClient: stepSimulation drives the clock (snake_gait_test.py:54)")

    def resetDebugVisualizerCamera(self, *a, **k):
        return None

    def getCameraImage(self, width, height, *a, **k):
        return (width, height, [], [], [])

    def getJointInfo(self, body, joint):
        """(index, name, type, ...) in PyBullet's layout, from the URDF's module pattern: joint 3k is
        module k's revolute joint (type 0), the others are fixed (type 4)."""
        if body != self._SNAKE or not (0 <= joint <= 3 * self._n):
            raise NotImplementedError("BulletClient.getJointInfo: the snake's joints 0 .. 3n")
        if joint == 0:
            name, link, k = "base_joint", "base", 0
        else:
            k = (joint + 2) // 3
            part = ("INPUT_INTERFACE", "COLLAR", "OUTPUT_BODY")[(joint - 1) % 3]
            name, link = "SA%03d_%s_joint" % (k, part), "SA%03d_%s" % (k, part)
        rev = joint >= 3 and joint % 3 == 0
        lo, hi = (self._world.get("joint_lo", -1.57), self._world.get("joint_hi", 1.57)) if rev else (0.0, -1.0)
        return (joint, name.encode(), 0 if rev else 4, (6 + k if rev else -1), (5 + k if rev else -1), 1,
                0.1 if rev else 0.0, 0.2 if rev else 0.0, lo, hi, 7.0 if rev else 0.0, 2.208932 if rev else 0.0,
                link.encode(), (0.0, 1.0, 0.0) if rev else (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), joint - 1)

    def enableJointForceTorqueSensor(self, body, joint, enableSensor=1):
        return None                                      # joints 0 and 3 are always evaluated (obs[55]; snake_gait_test.py:126)

    def getNumJoints(self, body):
        return 3 * self._n + 1 if body == self._SNAKE else 0

    # ---- the world itself ----
    def _stepper(self):
        if self._st is None:
            if self._SNAKE not in self._bodies:
                raise RuntimeError("BulletClient: no snake loaded (loadURDF) yet")
            self._st = _lib.Stepper(1, device=self._device, n_modules=self._n, **self._world)
            self._targets = np.zeros((1, self._n), dtype=np.float32)
            self._cache = None
        return self._st

    def _state(self):
        """(state, aux, link positions) of the world, fetched once per simulator state."""
        if self._cache is None:
            st = self._stepper()
            s, x = st.get_state()
            self._cache = (s[0].astype(np.float64), x[0].astype(np.float64), None)
        return self._cache

    def _motor(self, joint):
        if joint % 3 != 0 or not (3 <= joint <= 3 * self._n):
            raise NotImplementedError("BulletClient: joint %d is not a motor joint (3, 6, ..., 3n)" % joint)
        return joint // 3 - 1

    # ---- reset (snake.py:119-127) ----
    def resetBasePositionAndOrientation(self, body, posObj, ornObj):
        st = self._stepper()
        s, _ = st.get_state()
        s[0, 0:3] = posObj
        s[0, 3:7] = ornObj
        s[0, 7:13] = 0.0                                 # [U] zeroes the base twist
        st.set_state(s)
        self._cache = None

    def resetJointState(self, body, jointIndex, targetValue, targetVelocity=0.0):
        j = self._motor(jointIndex)
        st = self._stepper()
        s, _ = st.get_state()
        s[0, 13 + j] = targetValue
        s[0, 13 + self._n + j] = targetVelocity
        st.set_state(s)
        self._cache = None

    # ---- the substep (snake.py:219-221, 286) ----
    def _check_forces(self, forces):
        if forces is None:
            return
        f = np.asarray(forces, dtype=np.float64).reshape(-1)
        mi = self._world.get("max_motor_impulse", math.inf)
        dt = self._world.get("dt", 1.0 / 240.0)
        want = math.inf if np.all(np.isinf(f)) else float(f[0]) * dt
        if not np.all(f == f[0]):
            raise NotImplementedError("BulletClient: one force limit for all motors (snake.py:26-27)")
        if want != mi:
            self._set("max_motor_impulse", want)         # fine before the world exists, an error after

    def setJointMotorControlArray(self, bodyUniqueId, jointIndices, controlMode, targetPositions=None, forces=None, **kw):
        if controlMode != self.POSITION_CONTROL or kw:
            raise NotImplementedError("BulletClient: POSITION_CONTROL with PyBullet's default gains only (snake.py:221)")
        self._check_forces(forces)
        self._stepper()
        for j, t in zip(jointIndices, targetPositions):
            self._targets[0, self._motor(j)] = t

    def setJointMotorControl2(self, bodyIndex, jointIndex, controlMode, targetPosition=0.0, force=None, **kw):
        # snake_gait_test.py:71-76 drives the motors one by one (its positionGain / velocityGain are PyBullet's defaults
        # spelt out); snake.py:109-117's kp = 10 form is not on the path and not supported
        if controlMode != self.POSITION_CONTROL:
            raise NotImplementedError("BulletClient: POSITION_CONTROL only")
        if kw.get("positionGain", 0.1) != 0.1 or kw.get("velocityGain", 1.0) != 1.0:
            raise NotImplementedError("BulletClient: PyBullet's default gains only (kp 0.1, kd 1.0)")
        self._check_forces(None if force is None else [force])
        self._stepper()
        self._targets[0, self._motor(jointIndex)] = targetPosition

    def stepSimulation(self):
        self._stepper().substep(self._targets, 1)
        self._cache = None

    # ---- read-out (snake.py:130-146, 180-206, 237-245; snake_gait_test.py:33-40) ----
    def getJointState(self, bodyUniqueId, jointIndex):
        s, x, _ = self._state()
        n = self._n
        if jointIndex == 0:                              # the head sensor: obs[55] = reaction Fz (snake.py:202-206)
            return (0.0, 0.0, (0.0, 0.0, float(x[n]), 0.0, 0.0, 0.0), 0.0)
        j = self._motor(jointIndex)
        # the 6-D reaction is evaluated for joints 0 and 3 only (obs[55]; snake_gait_test.py:126 reads joint 3's Fz and
        # records the others without using them): the rest come back as NaN, not as a made-up zero
        nan = float("nan")
        react = (nan,) * 6
        if jointIndex == 3:
            react = (nan, nan, float(self._stepper().joint3_reaction_fz()[0]), nan, nan, nan)
        return (float(s[13 + j]), float(s[13 + n + j]), react, float(x[j]))

    def getBasePositionAndOrientation(self, bodyUniqueId):
        s, _, _ = self._state()
        return tuple(float(v) for v in s[0:3]), tuple(float(v) for v in s[3:7])

    def getLinkStates(self, bodyUniqueId, linkIndices, **kw):
        idx = [int(i) for i in linkIndices]
        if any(i % 3 != 0 or not (0 <= i <= 3 * self._n) for i in idx):
            raise NotImplementedError("BulletClient.getLinkStates: links 0, 3, ..., 3n (snake.py:142, 240)")
        s, x, lp = self._state()
        if lp is None:
            lp = self._stepper().link_positions()[0].astype(np.float64).reshape(3, self._n + 1)      # [x.., y.., z..]
            self._cache = (s, x, lp)
        # [0] = world position of the link's COM: the only field the path reads.  (PyBullet's further fields are 3- and
        # 4-vectors; snake.py:143 builds an array from the tuples, which numpy >= 1.24 only accepts when they have one
        # length -- hence a 3-vector placeholder in the orientation's place)
        return [((float(lp[0, i // 3]), float(lp[1, i // 3]), float(lp[2, i // 3])), (0.0, 0.0, 0.0)) for i in idx]
