for T in 1 2 4 8; do
  AMHIP_TUNING=session_upload_threads=$T python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-mode --no-rough-terrain 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);p=d['pcie_inclusive'];m=p['breakdown']['calls']['mosaic'];print('T=$T', p['ms'],p['dsm_ms'],'mosaic',m)"
done
