import sys, os
sys.path.insert(0, "/root/repo")
import torch; torch.cuda.init()
from halo2_gpu_specific_amd import prover, circuits
D = prover.Device()
dom = prover.Domain(22, 3)
print("plan", D.coset_plan(dom), dom.quotient_poly_degree, dom.extended_k, dom.k, D.group_size, D.force_cosets)
