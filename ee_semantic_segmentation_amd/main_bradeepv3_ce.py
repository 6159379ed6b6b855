#!/usr/bin/env python3
"""main_bradeepv3_ce.py: the cross-entropy variant (differs from main_bradeepv3.py only in
the loss, main_bradeepv3_ce.py:121)."""
from .main_bradeepv3 import main

if __name__ == "__main__":
    main("ce")
