# development aid: SPH kernels alone for several library builds (profiles/dev/libghip_<name>.so)
for v in "$@"; do
  cp profiles/dev/libghip_$v.so gadget-leicester_amd/libghip.so || exit 1
  echo "== $v"
  (cd tests && timeout -k 10 120 python gpu_sphperf.py 64 7) || exit 1
done
