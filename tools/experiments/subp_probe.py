import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess, sys
import numpy as np
from tensorbnn_amd import _native as nat
ch = nat.Chain([(3, 1, 0, 0)], kernel=nat.KERNEL_GENERIC)
ch.set_data(np.zeros((4, 3), np.float32), np.zeros((4, 1), np.float32))
print("gpu initialised:", ch.logp_grad()[0])
r = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True)
print("child rc", r.returncode, r.stdout.splitlines()[0] if r.stdout else r.stderr[:200])
import os
r = os.system("echo from-system; /opt/rocm/bin/hipcc --version | head -1")
print("system rc", r)
