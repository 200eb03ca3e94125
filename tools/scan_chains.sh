for B in 1 2 4 8 16 32 64; do
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --chains $B | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('chains $B value %.1f G/s'%(r['value']/1e9),'ms/step %.3f'%r['ms_per_step'],'OR launch %.3f ms'%r['roofline']['launch_ms'], 'frac %.3f'%r['roofline']['frac'],'HB %.3f ms'%r['heatbath']['launch_ms'])" || exit 1
done
