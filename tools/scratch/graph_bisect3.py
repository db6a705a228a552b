import sys, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_f3_scenarios as T
fn = T.test_graphed_step_replays_the_eager_step
fn = getattr(fn, "__wrapped__", fn)
name, fused = sys.argv[1], sys.argv[2] == "1"
fn(name, fused)
print(name, fused, "OK")
