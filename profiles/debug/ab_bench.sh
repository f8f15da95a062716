for i in 1 2; do for v in base new; do GTE_LIB_PATH=$PWD/_ab/libgte_$v.so python bench.py --no-cpu-baseline --no-gather-probe 2>&1 | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$v', round(d['ms_per_step'],4), {n: round(k[n]['avg_ms']*1e3,1) for n in ('narrow_fwd','narrow_bwd','spmm_csr') if n in k})"; done; done
