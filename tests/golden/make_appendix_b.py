"""Generates tests/golden/appendix_b.json: the URDF-derived known answers of SURVEY.md
Appendix B (rest-pose link COMs, joint axes, DFS motor indices, masses).

These are closed-form consequences of the constants in the reference's data file
snake/snake.urdf (cited per line in SURVEY.md Appendix B), written out here by hand --
no reference code is imported or executed and no PyBullet output is involved (PyBullet
is not installable here; parity with PyBullet itself stays UNPINNED).
"""
import json
import math
import os

N = 16
PITCH = 0.0366 + 0.0273          # snake.urdf:836,877
YAW = -1.57075                   # snake.urdf:877 (not -pi/2)

out = {
    "n_modules": N,
    "num_links_with_root": 3 * N + 2,
    "motor_joint_indices": list(range(3, 3 * N + 1, 3)),      # snake.py:80 arange(3, numJoints, 3)
    "height_sample_links": list(range(0, 3 * N + 1, 3)),      # snake.py:240 arange(0, numJoints, 3)
    "module_pitch": PITCH,
    "rest_mean_height": 0.026,
    "sum_declared_mass": 32 * 0.103,
    "num_links_without_inertial": 18,
    "total_mass_bullet_rule": 32 * 0.103 + 18 * 1.0,
    # Bullet link index 3k (OUTPUT_BODY k) COM at rest: x = -0.0366 - 0.0639 (k-1), y = 0, z = 0.026
    "output_body_com_rest": [[-0.0366 - PITCH * (k - 1), 0.0, 0.026] for k in range(1, N + 1)],
    "base_link_com_rest": [0.0, 0.0, 0.026],
    # world axis of motor slot k-1 at rest: Ry(-pi/2) Rz(YAW)^(k-1) e_y
    "joint_axes_rest": [],
}
for k in range(N):
    a = YAW * k
    # e_y rotated by Rz(a) in the base frame = (-sin a, cos a, 0); base x->world z, base y->world y, base z->world -x
    bx, by = -math.sin(a), math.cos(a)
    out["joint_axes_rest"].append([0.0, by, bx])

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "appendix_b.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path)
