"""cProfile of the host side of bench.py's timed loop (where does the Python time of a step go?).
    python tools/probes/host_profile.py --decoder att --batch 12 --steps 40"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py"] + sys.argv[1:] + ["--no-cpu-baseline", "--no-lstm-roofline", "--no-conv-events"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
sys.stderr.write(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
sys.stderr.write(s.getvalue())
