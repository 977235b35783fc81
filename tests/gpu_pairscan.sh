#!/bin/bash
# the step time against the Newtonian walk's wavefront cap inside a pair (GHIP_PAIR_NEWTON_LDS bytes of
# dynamic LDS per one-wavefront workgroup: 160 KB / value = workgroups per CU)
for so in "$@"; do
  for lds in ${LDS_LIST:-6826 8192 10240}; do
    echo -n "$so lds $lds : "
    GHIP_PAIR_NEWTON_LDS=$lds GHIP_LIBGHIP=$PWD/gadget-leicester_amd/variants/libghip_$so.so python bench.py --no-dropin --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['phases_ms_rank0']; print('%.3f ms/step  grav %.2f ewald %.2f dens %.2f hydro %.2f'%(d['ms_per_step'],p['grav'],p['ewald'],p['dens'],p['hydro']))"
  done
done
