import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from pvcnn_amd import workload
from pvcnn_amd.modules.functional import backend as seam

from oracle import oracle_backend; oracle_backend.build(); OracleBackend = oracle_backend.OracleBackend
oracle = OracleBackend()
for name in ('PVCNN', 'PVCNN2'):
    for seed in (7, 8, 9):
        torch.manual_seed(seed)
        cpu_net = getattr(workload, name)(13, 6, width_multiplier=0.25).eval()
        gpu_net = getattr(workload, name)(13, 6, width_multiplier=0.25)
        gpu_net.load_state_dict(cpu_net.state_dict()); gpu_net = gpu_net.to('cuda:0').eval()
        x, _ = workload.make_s3dis_batch(2, 2048, seed=seed)
        with torch.no_grad():
            got = gpu_net(x.to('cuda:0')).cpu()
            prev = seam._backend; seam._backend = oracle
            try: want = cpu_net(x)
            finally: seam._backend = prev
        d = ((got - want).abs() / (1 + want.abs()))
        print(name, seed, 'max', d.max().item(), 'frac>1e-4', (d > 1e-4).float().mean().item(), 'frac>1e-5', (d > 1e-5).float().mean().item(), 'scale', want.abs().max().item())
