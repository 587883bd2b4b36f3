#!/usr/bin/env python
"""fixed workload for `tools/gpurun.sh pmcsq flash tools/pmc_flash.py`: the 64x64 self-attention (N = 4096, d = 40, 8 heads), forward + backward"""
import os
import sys
sys.argv = [sys.argv[0], "--only", os.environ.get("CB_PMC_SHAPE", "self 64x64"), "--reps", "4", "--dtype", os.environ.get("CB_PMC_DTYPE", "f32"),
            "--batch", os.environ.get("CB_PMC_BATCH", "1")]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_flash
bench_flash.main()
