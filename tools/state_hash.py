"""Hash of the state and outputs after a few gait steps (to compare two builds bit for bit):
    SNK_LIB=<lib> python tools/state_hash.py [16|32]       one line per chain length
    python tools/state_hash.py --pins                      the block tests/test_gpu_bits.py pins (three kernel families, per part)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
pkg = importlib.import_module("bullet-envs_amd")
import test_gpu_bits as T

if "--pins" in sys.argv:
    key, txt = T.toolchain_key()
    print("    # %s" % txt.replace("\n", " || "))
    print('    "%s": {' % key)
    for n, streamed in ((16, False), (16, True), (32, False)):
        if streamed:
            os.environ["SNK_FORCE_STREAMED"] = "1"
        else:
            os.environ.pop("SNK_FORCE_STREAMED", None)
        print('        "%d%s": %r,' % (n, "s" if streamed else "", T.state_hashes(pkg, n, streamed)))
    print("    },")
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    print(n, "links:", T.state_hashes(pkg, n, bool(os.environ.get("SNK_FORCE_STREAMED")))["all"])
