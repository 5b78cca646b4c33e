"""Workloads of profiles/r6_binary.txt: binary erosion with the default structure on bool volumes (512^3 one iteration,
512^3 three fused iterations, 1024^3), each launched 10 times.  Run under scripts/kstat_any.sh / scripts/pmc_script.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi
for shape, its in [((512,) * 3, (1, 3)), ((1024,) * 3, (1,))]:
    b = ca.asarray(np.random.default_rng(0).random(shape) > 0.3); bo = ca.empty(shape, bool)
    for it in its:
        for _ in range(10): ndi.binary_erosion(b, iterations=it, output=bo)
        ca.synchronize()
    b = bo = None; ca.free_all_blocks()
