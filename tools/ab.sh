set -e
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG', 'step_us %.2f' % (1000*d['ms_per_step']), 'Mfps %.1f' % (d['value']/1e6), 'frac %.3f' % d['roofline']['frac'])"; }
variant() {
  TAG="$1"; export ASP_HIPCC_EXTRA="ns_kernels2.hip:$2"
  touch audiosignalprocess_amd/csrc/ns_kernels2.hip
  python -c "from audiosignalprocess_amd import build; build.build_library()"
  run; run; run
}
variant "perm0 order0" "-DNS_EXP_PERMLANE=0 -DNS_EXP_ORDER=0"
variant "perm1 order0" "-DNS_EXP_PERMLANE=1 -DNS_EXP_ORDER=0"
variant "perm0 order1" "-DNS_EXP_PERMLANE=0 -DNS_EXP_ORDER=1"
variant "perm1 order1" "-DNS_EXP_PERMLANE=1 -DNS_EXP_ORDER=1"
variant "perm0 order0 again" "-DNS_EXP_PERMLANE=0 -DNS_EXP_ORDER=0"
