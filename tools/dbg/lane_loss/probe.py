"""which lanes of delta-v does a build lose under the single-substep API?  From the rest pose every
joint is commanded to 0.2 rad; after ONE substep each joint has moved by kp x 0.2 = 0.02 (the position motor closes 10 %
of the error).  Prints q after one substep, joint by joint.  SNK_LIB selects the library."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bullet-envs_amd")
st = pkg.Stepper(2)
st.reset()
T = np.full((2, 16), 0.2, np.float32)
st.substep(T, 1)
S, _ = st.get_state()
print(os.environ.get("SNK_LIB", "libsnk.so"), "q after one substep:", np.round(S[0, 13:29], 4))
print("  qd:", np.round(S[0, 29:45], 3))
st.close()
