import os, sys, subprocess, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import bindings as ob
import gguf_synth as gs
ref = ob.Ref()
p = gs.tiny_model('/tmp/dense.gguf', ref, n_vocab=512)
pm = gs.tiny_model('/tmp/moe.gguf', ref, n_vocab=512, n_expert=4, n_used=2, seed=2)
L = 'oracle/_ref/llama/bin/llama_logits'
for model in (p, pm):
    for kv in (None, '1'):
        e = dict(os.environ); e['GGML_CDNA4_LOG_UNSUPPORTED'] = '1'
        if kv: e['LLAMA_LOGITS_KV_OFFLOAD'] = '1'
        r = subprocess.run([L, model, '99', '48', '8', 'none', '/tmp/o.bin', '2'], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        print('=====', model, 'kv offload', kv, 'rc', r.returncode)
        print('\n'.join(sorted(set(l for l in r.stderr.decode().split('\n') if l.startswith('cdna4-unsupported')))))
        print('\n'.join(l for l in r.stderr.decode().split('\n') if 'flash' in l or 'graph splits' in l))
