#!/usr/bin/env python3
"""Per-kernel step times of one libchub build, for A/B comparisons INSIDE one GPU-box call (box-to-box variation is
+-3 %, far more than most kernel changes):  for v in a b a b; do CHUB_LIB=$PWD/charginghub-env_amd/libchub_$v.so python
tools/ab_step_times.py; done"""
import os, sys
sys.path.insert(0, ".")
import torch
import charginghub_env_amd as chub
n=65536
for perm in (0.01,):
    kw = dict(station_list=[20,25], station_type_list=["fast","slow"], hydro_prod_rate=100.0, hydro_store_vlt=25.0, init_soc=0.2, fc_max_power=100.0, fcev_permeate=perm)
    v = chub.VecChargingHub(n, seed=1, **kw)
    dev = torch.device("cuda", 0)
    acts = [torch.empty((n, 47), device=dev) for _ in range(4)]
    for b,a in enumerate(acts): v.random_actions_device(a.data_ptr(), 123, b, 0)
    packed = torch.empty((n, 15), device=dev); obs0 = torch.empty((n, 13), device=dev)
    for i in range(960):
        if i % 96 == 0: v.reset_device(obs0.data_ptr())
        v.step_device_packed(acts[i%4].data_ptr(), packed.data_ptr())
    torch.cuda.synchronize()
    v.profile_begin(1920, every=2)
    for i in range(1920):
        if i % 96 == 0: v.reset_device(obs0.data_ptr())
        v.step_device_packed(acts[i%4].data_ptr(), packed.data_ptr())
    a,b,k = v.profile_end()
    print(os.environ.get("CHUB_LIB","")[-12:], "permeate", perm, "slot_us %.2f env_us %.2f" % (a/k*1e3, b/k*1e3))
    v.close()
