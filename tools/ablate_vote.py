import sys; sys.path.insert(0,'.')
from localhgt_amd.engine import Engine
k,e=32,3
eng=Engine(k,e); eng.rng_seed(1); eng.coder_generate()
NC,CL,NP=1000,1_000_000,10_000_000
eng.synth_reference(1,NC,CL); eng.synth_pairs(1,2,NC,CL,0,NP)
eng.count_kmers(); n=eng.ref_scan(0.1,0.08,3*10**8); print("peaks",n, "A ms", eng.phase_ms(0), "B ms", eng.phase_ms(1))
for flags in (0,1,0,1):
    eng.set_debug(flags); eng.vote(); print("debug",flags,"vote ms",eng.phase_ms(2))
