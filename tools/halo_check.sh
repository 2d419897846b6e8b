cd /root/repo
for bs in "8 1" "4 2" "4 1" "2 2" "16 1"; do set -- $bs; timeout 300 python bench.py --size 3840x2160 --batch $1 --streams $2 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4K batch $1 streams $2', round(d['value'],1))"; done
