import sys, time, os
sys.path.insert(0,''+os.path.dirname(os.path.dirname(os.path.abspath(__file__)))+''); sys.path.insert(0,''+os.path.dirname(os.path.dirname(os.path.abspath(__file__)))+'/tests')
import numpy as np
import dnastore_amd as da
from oracle import oracle as O
from synth import synthetic_reads
G=''+os.path.dirname(os.path.dirname(os.path.abspath(__file__)))+'/tests/golden/ref_data/'
m=da.Machine.fromFile(G+'s16h74l4c4.json'); p=da.MutatorParams.fromFlags(global_=True)
om=O.Machine.from_file(G+'s16h74l4c4.json'); orc=O.ViterbiOracle(om,O.MutatorParams.from_cli(global_=True))
dec=da.ViterbiDecoder(m,p)
reads=synthetic_reads(om,64,29,seed=1000,sub=0.01)
print('lens',[len(r) for r in reads[:8]])
for nb in (1,8,64):
    t=time.time(); out,ll,st=dec.decode(reads[:nb]); dt=time.time()-t
    s=dec.stats(); nt=sum(len(r) for r in reads[:nb])
    print(nb,'reads wall %.3fs'%dt, s, 'nt/s(fill) %.0f'%(nt/(s['fill_ms']/1e3)), 'rounds/col %.1f'%(s['rounds']/s['columns']), 'GB/s %.1f'%(s['lattice_bytes']/s['fill_ms']/1e6))
t=time.time(); s0,l0=orc.decode(reads[0]); print('oracle %.2fs'%(time.time()-t), s0==out[0], l0==ll[0], l0, ll[0], st[:4])
