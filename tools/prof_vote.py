import cProfile, pstats, sys, os, io
sys.argv = ["bench.py"] + sys.argv[1:]
sys.path.insert(0, os.getcwd())
import bench
from scd_amd import naming
orig_u, orig_p = naming.vote_loop_unsup, naming.vote_loop_ptsup
pr = cProfile.Profile()
def wrap(f):
    def g(*a, **k):
        pr.enable()
        try:
            return f(*a, **k)
        finally:
            pr.disable()
    return g
naming.vote_loop_unsup = wrap(orig_u); naming.vote_loop_ptsup = wrap(orig_p)
import scd_amd.pipeline as pl
bench.main()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:5000], file=sys.stderr)
