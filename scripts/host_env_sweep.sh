mkdir -p gpurun_out/r05
run() { name=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r05/env_$name.json 2> gpurun_out/r05/env_$name.err; python -c "
import json; d=json.load(open('gpurun_out/r05/env_$name.json')); print('%-28s img/s %.0f ms %.2f host_issue med %.1f min %.1f cpu med %.1f' % ('$name', d['value'], d['ms_per_step'], d['host_issue_ms_median'], d['host_issue_ms_min'], d['host_cpu_ms_median']))" || tail -3 gpurun_out/r05/env_$name.err; }
run base X=1
run kernarg16M HSA_KERNARG_POOL_SIZE=16777216
run aql64k ROC_AQL_QUEUE_SIZE=65536
run activewait0 ROC_ACTIVE_WAIT_TIMEOUT=0
run cpuwait ROC_CPU_WAIT_FOR_SIGNAL=1
run signalpool ROC_SIGNAL_POOL_SIZE=4096
run all HSA_KERNARG_POOL_SIZE=16777216 ROC_AQL_QUEUE_SIZE=65536 ROC_SIGNAL_POOL_SIZE=4096
