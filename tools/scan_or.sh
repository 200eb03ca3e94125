for nt in 256 512 1024; do for f in 1 2 3 4; do
MLMCPI_OR_THREADS=$nt timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --fuse $f --n-overrelax 12 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('NT $nt fuse',r['config']['fuse'],'value %.1f G/s'%(r['value']/1e9),'OR launch %.3f ms'%r['roofline']['launch_ms'],'per sweep %.3f'%(r['roofline']['launch_ms']/$f), 'frac %.3f'%r['roofline']['frac'],'HB %.2f ms'%r['heatbath']['launch_ms'])" || exit 1
done; done
