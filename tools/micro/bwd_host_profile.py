import cProfile, pstats, sys, os, runpy
sys.argv = ["tools/time_backward_step.py", "--reps", "3"]
cProfile.run("runpy.run_path('tools/time_backward_step.py', run_name='__main__')", "/tmp/bwd.prof")
p = pstats.Stats("/tmp/bwd.prof"); p.sort_stats("tottime").print_stats(18)
