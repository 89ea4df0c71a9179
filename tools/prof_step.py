"""cProfile of the HOST side of bench.py's steps (pipeline.run / run_ptsup / run_cached), warm-up included: where the Python between the
launches goes.  python tools/prof_step.py [bench.py args]  -> the profile on stderr, the bench line on stdout."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"] + sys.argv[1:]
import bench
import scd_amd.pipeline as pl
pr = cProfile.Profile()


def wrap(f):
    def g(*a, **k):
        pr.enable()
        try:
            return f(*a, **k)
        finally:
            pr.disable()
    return g


for name in ("run", "run_ptsup", "run_cached"):
    setattr(pl, name, wrap(getattr(pl, name)))
bench.main()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:7000], file=sys.stderr)
