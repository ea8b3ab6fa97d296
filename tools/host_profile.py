"""cProfile of the host side of EM steps on a tiny batch (the GPU is idle most of the time)."""
import cProfile, pstats, sys, os
sys.argv = ["bench.py", "--steps", "300", "--warmup", "20", "--batch", "100", "--no-cpu-baseline", "--no-kernel-timing"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
