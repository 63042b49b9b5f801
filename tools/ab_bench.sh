# A/B of library variants / env knobs on one bench workload (WORKLOAD=cfg3 by default), same box, interleaved rounds: tools/ab_bench.sh "<name>[:ENV=VAL...]" ...
# name "" = the default library; other names = airwave_amd/libairwave_hip_<name>.so
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=""; [ "$spec" != "$name" ] && envs=${spec#*:}
  lib=""; [ -n "$name" ] && [ "$name" != "-" ] && lib=$GRAFT_REPO_ROOT/airwave_amd/libairwave_hip_$name.so
  echo -n "[$r] $spec: "
  env AIRWAVE_HIP_LIBRARY=$lib $(echo $envs | tr ',' ' ') python bench.py --workload ${WORKLOAD:-cfg3} --no-cpu-baseline --no-secondary --steps ${STEPS:-5} --warmup ${WARMUP:-1} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print(round(d['value']/1e9,2),'G/s', round(d['ms_per_step'],2),'ms', {k.replace('aw_part_','').replace('_kernel',''):round(v,2) for k,v in r['stages_ms_per_step'].items()})"
done
done
