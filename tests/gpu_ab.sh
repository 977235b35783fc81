# development aid: A/B of two library builds on the same box (profiles/dev/libghip_<name>.so)
for v in "$@"; do
  name=${v%%:*}; env=${v#*:}; [ "$env" = "$v" ] && env=""
  cp profiles/dev/libghip_$name.so gadget-leicester_amd/libghip.so || exit 1
  echo "== $name $env"
  (cd tests && env $env timeout -k 10 200 python gpu_ewald_alone.py 64) || exit 1
  env $env timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-dropin 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['phases_ms_rank0'])" || exit 1
done
