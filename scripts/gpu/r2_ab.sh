#!/bin/bash
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
timeout 1500 python scripts/gpu/ab_step.py 2 "base:" "mainprio:PPF_MAIN_PRIORITY=1"
