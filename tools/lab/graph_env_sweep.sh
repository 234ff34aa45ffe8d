#!/bin/bash
# hipGraph replay vs eager under the runtime's graph switches: ms/step and host issue time per step
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
out=gpurun_out/graph_env_sweep.txt; : > $out
one() {  # env-assignments -- bench args
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-dense-reference --steps 10 "$@" 2> gpurun_out/sweep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d.get('eager') or {}
print('%-45s %-40s graph %.2f ms (host %.2f) | eager %.2f ms (host %.2f) | plain %.2f' % ('${envs[*]}', '$*', d['ms_per_step'], d['host_issue_ms_per_step'], e.get('ms_per_step',0), e.get('host_issue_ms_per_step',0), (d.get('lookahead') or {}).get('ms_per_step_without',0)))
" >> $out 2>&1 || { echo "FAILED ${envs[*]} $*" >> $out; tail -3 gpurun_out/sweep.err >> $out; }
}
for e in X=1; do
  one $e -- --workload model --clouds 2
  one $e -- --workload fixmatch --points 16000
done
one X=1 -- --workload model --clouds 1
one X=1 -- --workload model --clouds 4
one X=1 -- --workload model
one X=1 -- --workload fixmatch
cat $out
