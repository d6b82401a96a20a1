"""The training-step measurement of bench.py (`train_step`) on its own: N_RAND rays per step."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import train_step_rate

if __name__ == "__main__":
    print(json.dumps(train_step_rate("cuda:0", int(os.environ.get("N_RAND", "4096")))))
