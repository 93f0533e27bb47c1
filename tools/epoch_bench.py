"""bench.training_epochs alone: ms per epoch of Predictor.train and where it goes (host marks partition the loop's wall time;
device time of the steps / the validation pass from events)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
r = bench.training_epochs(torch.device("cuda", 0), int(sys.argv[1]) if len(sys.argv) > 1 else 60)
for k, v in r.items():
    print(k, json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a not in ("host_ms_per_epoch", "note")}))
    print("    host ms per epoch:", json.dumps({a: round(b, 4) for a, b in v["host_ms_per_epoch"].items()}))
