#!/usr/bin/env python3
"""Developer tool: scripts/stress.py restricted to the shapes the 120-variable variants solve (standing, mixed, h = 20 single)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "scripts", "stress.py")).read()
src = src.replace('cases = [("standing", 10, 2), ("walking", 10, 2), ("mixed", 10, 2), ("single", 20, 2), ("standing", 16, 2), ("standing", 20, 2), ("3contact", 10, 3)]',
                  'cases = [("standing", 10, 2), ("mixed", 10, 2), ("single", 20, 2)]')
src = src.replace('ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))', 'ROOT = %r' % ROOT)
sys.argv = ["stress.py"] + sys.argv[1:]
exec(compile(src, "stress_120", "exec"))
