for W in 28 40; do
timeout 900 python bench.py --workload fw --fw-max-width $W --steps 25 --warmup 0 --pmc 0 --cpu-sample 512 --validate 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['config']; print('fw $W', '%.3e'%j['value'], 'bad replicas', c.get('validated_bad_replicas'), 'cpu sample bit-exact', c.get('cpu_sample_min_cost_bit_exact'), 'best', c.get('best_log10_flops'), j['roofline'].get('reslices'))"
done
timeout 900 python bench.py --workload im --steps 25 --warmup 0 --pmc 0 --cpu-sample 2048 --validate 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['config']; print('im', '%.3e'%j['value'], 'bad replicas', c.get('validated_bad_replicas'), 'cpu sample bit-exact', c.get('cpu_sample_min_cost_bit_exact'), 'best', c.get('best_log10_flops'))"
timeout 900 python bench.py --workload im --replicas 524288 --steps 3 --warmup 1 --pmc 0 --cpu-sample 64 --validate 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['config']; print('im 524288', '%.3e'%j['value'], j['ms_per_step'], 'bad replicas', c.get('validated_bad_replicas'), 'cpu sample bit-exact', c.get('cpu_sample_min_cost_bit_exact'))"
