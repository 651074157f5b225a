import numpy as np, sys
sys.path.insert(0,'.')
from localhgt_amd.engine import Engine
k,e=32,3
eng=Engine(k,e); eng.rng_seed(1); eng.coder_generate()
NC,CL,NP=20,1_000_000,200_000
eng.synth_reference(1,NC,CL); eng.synth_pairs(1,2,NC,CL,0,NP)
eng.count_kmers(); n=eng.ref_scan(0.1,0.08,10**8)
fl=eng.flags_export(0,NC*CL)
print("peaks",n,"selected",int(((fl>>5)&1).sum()),"peakflag",int(((fl>>3)&1).sum()),"inside",int(((fl>>4)&1).sum()),"single",int((fl&1).sum()))
pk=eng.peak_kmer_export(0,1<<28)
print("peak_kmer nonzero frac (first 2^28 slots)", (pk!=0).mean(), (pk!=0).sum())
print("hist", eng.counts_histogram())
eng.vote(); print("ms", eng.phase_ms(0), eng.phase_ms(1), eng.phase_ms(2))
