#!/usr/bin/env python3
"""Command-line entry with the flags of the reference's demo.py
(--config --checkpoint --image | --image_dir --instruction --output --device),
running the MI355X forward pass with a real batch loop; see
cognitive_aim_depth_estimation_amd/inference.py.

Multi-GPU (image-sharded, one process per GPU):
    python -m torch.distributed.run --nproc-per-node 8 demo.py --image_dir photos/ ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from cognitive_aim_depth_estimation_amd.inference import main  # noqa: E402

if __name__ == "__main__":
    main()
