import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sys
from tensorbnn_amd import jit
from tensorbnn_amd.workloads import synth_problem
dims = [int(v) for v in sys.argv[1].split(",")]
layers, lik, X, Y, th, eta = synth_problem(dims, 1000)
print(jit.build(layers, lik, verbose=True))
