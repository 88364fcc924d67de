"""cProfile of the reference-shaped training loop on the GPU box (host-side cost per env step)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nlbac_amd  # noqa: F401
from nlbac_amd import train
argv = ["--env", "Unicycle", "--gamma_b", "50", "--max_episodes", "3", "--cuda", "--updates_per_step", "2",
        "--batch_size", "128", "--seed", "0", "--start_steps", "1000", "--device_replay", "--max_steps", "1800"] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
train.main(argv)
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("cumulative").print_stats(24)
st.sort_stats("tottime").print_stats(18)
print(s.getvalue()[:9000])
