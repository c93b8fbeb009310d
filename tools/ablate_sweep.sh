for a in ${ABLATES:-0 8 1 2 3}; do SPEECH_ABLATE=$a timeout -k 10 100 python bench.py --no-extras --steps 20 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('ablate $a', 'sustained', round(r['kernel_ms_avg'],4), 'isolated', round(r['kernel_ms_isolated_avg'],4), round(r['kernel_ms_isolated_min'],4))" || exit 1; done
