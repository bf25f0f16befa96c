#!/bin/bash
# What is the Hogwild lag made of?  C2, 32 epochs, device / sequential oracle under variations of the hub threshold, the stale budget and the worker count.
O=gpurun_out/r03/lag; mkdir -p $O
for V in "" "hot_theta=0.05" "hot_theta=0.01" "stale_budget=500.0" "stale_budget=8000.0" "workers=1024" "workers=256" "hot_theta=0.05,stale_budget=500.0"; do
  T=$(echo "${V:-default}" | tr ',=' '__')
  python3 tools/r03/convergence.py device --epochs 32 --ref profiles/r03_convergence_oracle.npz --device-cfg "$V" --out $O/lag_$T.json > $O/lag_$T.txt 2>&1
  python3 -c "
import json;d=json.load(open('$O/lag_$T.json'));r=d['device_over_oracle'];print('%-36s workers %5d  ratio @1 %.3f @5 %.3f @10 %.3f @20 %.3f @32 %.3f' % ('${V:-default}', d['workers'], r[0], r[4], r[9], r[19], r[31]))"
done
