#!/usr/bin/env python3
"""three more [U] rules priced in the oracle alone -- what each does to what a trainer sees under the
bench gait (32 envs x 200 env-steps, auto-reset on, float64, the same driver as tests/test_gpu_contact_models.py::
test_contact_model_error_bar and profiles/r03_contact_models.json):

  friction_directions 1    Bullet's multibody solver without SOLVER_USE_2_FRICTION_DIRECTIONS: one btPlaneSpace1 row per
                           contact, box bounds, no cone branch
  warm_start 1 x 0.1       warm starting with the factor PyBullet's world constructor is read as setting, beside
                           btContactSolverInfo's 0.85
  contact_erp_rule 1       the ERP of a contact row chosen by the split-impulse penetration threshold (m_erp 0.2 for
                           anything shallower than 4 cm) instead of m_erp2 0.08 throughout

Round 6 adds the two rules about ROW ORDER:

  noncontact_order 1       the non-contact rows in the order btAlignedObjectArray::quickSort leaves the world's constraint list
                           [limit_1..limit_16, motor_1..motor_16] in (equal island ids; Hoare partition, not stable): the motors
                           first, as 5 4 7 6 1 0 3 2 13 12 15 14 9 8 11 10, then the limits
  contact_order 1          the ground manifolds reversed
  contact_order 2          ... in link order after the island manager's quickSort on equal island ids (the same unstable sort,
                           over the 32 plane-link manifolds: the one candidate that can be restated)
  contact_order 3, 4, 5    ... in three fixed pseudo-random orders

Writes profiles/r06_u_rows.json (all rows, round 5's included); prints the table.  CPU only."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import oracle as orc  # noqa: E402

orc.build()
B, T = 32, 200
ids = np.arange(B)
threads = min(16, len(os.sched_getaffinity(0)))
ROWS = [
    ("default (Bullet's model as restated)", {}),
    ("friction_directions 1", dict(friction_directions=1)),
    ("warm_start 1, factor 0.85", dict(warm_start=1)),
    ("warm_start 1, factor 0.1", dict(warm_start=1, warmstarting_factor=0.1)),
    ("contact_erp_rule 1", dict(contact_erp_rule=1)),
    ("contact_erp_rule 1 + warm_start 1 x 0.1", dict(contact_erp_rule=1, warm_start=1, warmstarting_factor=0.1)),
    ("cone_friction 0 (pyramid, two directions)", dict(cone_friction=0)),
    ("noncontact_order 1 (quickSort on equal island ids)", dict(noncontact_order=1)),
    ("contact_order 1 (manifolds reversed)", dict(contact_order=1)),
    ("contact_order 2 (link order after Bullet's quickSort on equal island ids)", dict(contact_order=2)),
    ("contact_order 3 (fixed permutation A)", dict(contact_order=3)),
    ("contact_order 4 (fixed permutation B)", dict(contact_order=4)),
    ("contact_order 5 (fixed permutation C)", dict(contact_order=5)),
    ("noncontact_order 1 + contact_order 2 (both quickSorts)", dict(noncontact_order=1, contact_order=2)),
    ("noncontact_order 1 + contact_order 1", dict(noncontact_order=1, contact_order=1)),
    # context, not a [U] rule (Bullet's world runs 50 iterations): where the two orders go when the solve is allowed to converge
    ("n_iterations 200, link order", dict(n_iterations=200)),
    ("n_iterations 200, contact_order 2", dict(n_iterations=200, contact_order=2)),
    ("n_iterations 1000, link order", dict(n_iterations=1000)),
    ("n_iterations 1000, contact_order 2", dict(n_iterations=1000, contact_order=2)),
]
out = {}
base = None
print("%-76s %9s %8s %10s %10s %9s %8s" % ("switch", "substeps", "ends", "reward", "dx", "contacts", "dx vs default"))
for name, over in ROWS:
    _, _, agg = orc.bench_gait(B, bench.env_phases(ids), 0, T, threads, want_agg=True, max_contacts=0, **over)
    agg = {k: float(v) for k, v in agg.items()}
    if base is None:
        base = agg
    agg["dx_rel_to_default"] = agg["mean_dx"] / base["mean_dx"] - 1.0
    out[name] = dict(switches=over, oracle=agg)
    print("%-76s %9.3f %8.4f %10.5f %10.6f %9.1f %+7.1f %%" % (name, agg["mean_substeps"], agg["episode_end_rate"], agg["mean_reward"],
                                                            agg["mean_dx"], agg["mean_contacts"], 100 * agg["dx_rel_to_default"]))
with open(os.path.join(ROOT, "profiles", "r06_u_rows.json"), "w") as f:
    json.dump(dict(workload="bench gait, %d envs x %d env-steps, float64 oracle, uncapped contacts" % (B, T), rows=out), f, indent=1)
