for args in "--arms 3" "--arms 5" "--arms 3 --genes 5032" "--rehearse-dp"; do
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-eval $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,3), 'M cells/s; bf16', round(d.get('bf16_config',{}).get('ms_per_step',0),4))"
done
