#!/usr/bin/env python3
"""Same contract as the reference's scripts/get_bed_file.py: get_bed_file.py <ref.fa> <interval.txt>."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from localhgt_amd.get_bed_file import main
sys.exit(main())
