cd $GRAFT_REPO_ROOT
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1 value %.4g median %.4f min %.4f events %s clock %s form %s cp %s'%(d['value'], d['timed_region_ms']['median'], d['timed_region_ms']['min'], r['avg_launch_ms'], r.get('avg_launch_ms_device_clock'), r.get('rate_stage_form'), d['config'].get('control_plane')))
"; }
A="--gpus 1 --steps 256 --warmup 32 --no-cpu-baseline --no-secondary"
for i in 1 2; do
python bench.py $A 2>/dev/null | summ plain
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --force-process-group $A 2>/dev/null | summ ranked
done
A="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
python bench.py $A 2>/dev/null | summ plain20
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --force-process-group $A 2>/dev/null | summ ranked20
