"""snake(N) as URDF text, generated from the model constants -- TEST INFRASTRUCTURE.

Why: the arithmetic of the hot path lives in the third-party `pybullet` wheel, which is absent from this image
(parity unpinned).  Should a PyBullet ever be importable on a box these tests run on,
oracle/pybullet_live.py loads THIS text with the reference's own call sequence (snake.py:88-107) and compares
PyBullet's substeps with the oracle's -- the pinning plan of SURVEY.md Appendix C-1.  The text is produced from
the numbers below, it is not a copy of the reference's snake/snake.urdf (whose <gazebo>, <transmission> and
<visual> blocks Bullet ignores on this path anyway).

Structure per module k = 1..n (what fixes Bullet's DFS link indices, snake.py:80 `motorList = arange(3, 49, 3)`):
    INPUT_IF_k  --fixed-->  COLLAR_k                      (declared first: index 3k-1)
    INPUT_IF_k  --revolute (0,0,0.0366), axis y-->  OUTPUT_BODY_k      (index 3k)
    OUTPUT_BODY_k  --fixed (0,0,0.0273) rpy (0,0,-1.57075)-->  INPUT_IF_k+1   (index 3k+1)
in front of them: kdl_dummy_root --fixed (0,0,0.026) rpy (0,-pi/2,0)--> base --fixed--> INPUT_IF_1.
"""

CYL_R, CYL_L, CYL_Z = 0.026, 0.033, 0.0183           # urdf:806-811, 862-867
MASS, IXX, IZZ = 0.103, 5.4796e-5, 3.4814e-5          # urdf:812-816, 868-872
PIVOT_Z, NEXT_Z, NEXT_YAW = 0.0366, 0.0273, -1.57075  # urdf:833-840, 874-878
ROOT_Z, ROOT_PITCH = 0.026, -1.57079632679            # urdf:8-12
JOINT = dict(damping=0.1, friction=0.2, effort=7.0, lower=-1.57, upper=1.57, velocity=2.208932)   # urdf:838-839


def _inertial(z):
    return ('    <inertial>\n      <origin xyz="0 0 %r" rpy="0 0 0"/>\n      <mass value="%r"/>\n'
            '      <inertia ixx="%r" ixy="0" ixz="0" iyy="%r" iyz="0" izz="%r"/>\n    </inertial>\n'
            % (z, MASS, IXX, IXX, IZZ))


def _cylinder():
    return ('    <collision>\n      <origin xyz="0 0 %r" rpy="0 0 0"/>\n      <geometry><cylinder radius="%r" length="%r"/></geometry>\n'
            '    </collision>\n' % (CYL_Z, CYL_R, CYL_L))


def _link(name, body=""):
    return '  <link name="%s">\n%s  </link>\n' % (name, body) if body else '  <link name="%s"/>\n' % name


def _fixed(name, parent, child, xyz=(0, 0, 0), rpy=(0, 0, 0)):
    return ('  <joint name="%s" type="fixed">\n    <parent link="%s"/>\n    <child link="%s"/>\n'
            '    <origin xyz="%r %r %r" rpy="%r %r %r"/>\n  </joint>\n' % ((name, parent, child) + tuple(xyz) + tuple(rpy)))


def snake_urdf(n=16):
    out = ['<?xml version="1.0"?>\n<robot name="snake%d">\n' % n]
    out.append(_link("kdl_dummy_root"))
    out.append(_fixed("kdl_dummy_root_to_base", "kdl_dummy_root", "base", (0, 0, ROOT_Z), (0, ROOT_PITCH, 0)))
    out.append(_link("base"))
    out.append(_fixed("head__OUTPUT_INTERFACE", "base", "SA001__MoJo__INPUT_INTERFACE"))
    for k in range(1, n + 1):
        p = "SA%03d__MoJo__" % k
        out.append(_link(p + "INPUT_INTERFACE", _cylinder() + _inertial(PIVOT_Z)))
        out.append(_link(p + "INPUT_INTERFACE__COLLAR"))
        out.append(_fixed(p + "INPUT_INTERFACE__COLLAR_JOINT", p + "INPUT_INTERFACE", p + "INPUT_INTERFACE__COLLAR"))
        out.append('  <joint name="%s" type="revolute">\n    <parent link="%sINPUT_INTERFACE"/>\n'
                   '    <child link="%sOUTPUT_BODY"/>\n    <origin xyz="0 0 %r" rpy="0 0 0"/>\n    <axis xyz="0 1 0"/>\n'
                   '    <dynamics damping="%r" friction="%r"/>\n'
                   '    <limit effort="%r" lower="%r" upper="%r" velocity="%r"/>\n  </joint>\n'
                   % (p[:-2], p, p, PIVOT_Z, JOINT["damping"], JOINT["friction"], JOINT["effort"], JOINT["lower"],
                      JOINT["upper"], JOINT["velocity"]))
        out.append(_link(p + "OUTPUT_BODY", _cylinder() + _inertial(0.0)))
        if k < n:
            out.append(_fixed(p + "OUTPUT_INTERFACE", p + "OUTPUT_BODY", "SA%03d__MoJo__INPUT_INTERFACE" % (k + 1),
                              (0, 0, NEXT_Z), (0, 0, NEXT_YAW)))
    out.append("</robot>\n")
    return "".join(out)


def plane_urdf():
    """pybullet_data/plane.urdf as this path uses it: a static plane z = 0, lateral friction 1 [U]."""
    return ('<?xml version="1.0"?>\n<robot name="plane">\n  <link name="planeLink">\n'
            '    <contact><lateral_friction value="1"/></contact>\n'
            '    <inertial><origin xyz="0 0 0"/><mass value="0"/><inertia ixx="0" ixy="0" ixz="0" iyy="0" iyz="0" izz="0"/></inertial>\n'
            '    <collision><origin xyz="0 0 0"/><geometry><plane normal="0 0 1"/></geometry></collision>\n'
            '  </link>\n</robot>\n')


if __name__ == "__main__":
    import sys
    sys.stdout.write(snake_urdf(int(sys.argv[1]) if len(sys.argv) > 1 else 16))
