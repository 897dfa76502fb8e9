#!/usr/bin/env python3
"""Planning time (K0 + K1 + K1c in front of the first K2 of a call that follows a synchronisation) against the number of blocks per
window, small and large engines (scripts/k1_probe.py's scene)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); os.chdir(ROOT)
src = open(os.path.join(ROOT, "scripts", "k1_probe.py")).read().split('run("idle voices')[0].replace("ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", "")
exec(compile(src, "k1_probe_head", "exec"))
run("96 voices N=256 K=2048 ", V=96, B=12, N=256, KB=2048)
run("96 voices N=128 K=4096 ", V=96, B=12, N=128, KB=4096)
run("96 voices N=64  K=8192 ", V=96, B=12, N=64, KB=8192)
run("96 voices N=64  K=2048 ", V=96, B=12, N=64, KB=2048)
run("1024 voices N=64 K=8192", V=1024, B=8, N=64, KB=8192)
