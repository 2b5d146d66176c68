import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
args = bench.parse_args(sys.argv[1:])
if not args.no_numa:
    from f1tenth_gym_amd import numa
    print("numa:", numa.bind_to_device(0))
r = bench.dropin_rates(args)
print("f110env us", round(r['f110env_1env']['us_per_step'], 1))
for k in ('vecenv_2048', 'vecenv_32768'):
    print(k, {kk: round(v['ms_per_step'], 4) for kk, v in r[k].items() if isinstance(v, dict)})
