"""cProfile of the model step on a HOST state (development tool, GPU): where the 35 ms per step go."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import logging; logging.disable(logging.WARNING)
import device_step_timing as d
d.loop(False, 3)
pr = cProfile.Profile(); pr.enable()
print(d.loop(False, 10))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
