for i in 1 2; do for lib in flingbot_amd/libflingsim.so variants/libfs_pc2.so variants/libfs_pc6.so variants/libfs_st3.so variants/libfs_st6.so; do FLINGSIM_LIB=$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-eval-loop 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib', d['value'], d['ms_per_step'], d.get('parity',{}).get('bit_exact'))
"; done; done
