#!/usr/bin/env python
"""Shim with the reference's entry-point path: ``python src/experiment.py -c config.yaml -e econfigs/X.yaml``."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from deep_cbrs_amar_renaissance_amd.experiment import main  # noqa: E402

if __name__ == "__main__":
    main()
