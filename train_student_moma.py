#!/usr/bin/env python3
"""Entry point with the reference's script name: `python train_student_moma.py --distill moma ...`
(see moma_amd/train_student_moma.py for the flags)."""
from moma_amd.train_student_moma import main

if __name__ == "__main__":
    main()
