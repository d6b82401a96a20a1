"""Time of one training step (bench.py train_step_rate) in both training precisions, for profiling runs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import train_step_rate
for prec in sys.argv[1:] or ["bf16", "fp32"]:
    r = train_step_rate(torch.device("cuda:0"), steps=7, precision=prec)
    print(prec, json.dumps({k: v for k, v in r.items() if k != "what"}))
