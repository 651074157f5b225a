"""The other regimes of the same path (N = 1), a few steps each: results a reader needs next to the headline, whose synthetic
sample (half of a 13 Gbase reference) saturates the 2^32-slot table and yields no voted peak.  Order: what the compact line needs
first (the found-something workload), then by cost; the line is printed again after each group."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

from .legs import e2e_from_files, leg, pipelined_samples, reroof
from .pmc import PMC_PASSES, collect_pmc, pmc_traffic


def run_all(detail, eng, args, wl, local, emit):
    from localhgt_amd.engine import Engine
    out = detail.setdefault("secondary", {})
    k, e, L = args.k, args.e, 150
    nc, cl = args.contigs, args.contig_len
    headline = (args.contigs, args.pairs, args.sample_contigs, args.ragged, args.snp) == (13000, 100_000_000, 0, False, 0) and (k, e) == (32, 3)
    packed = args.ref_form == "packed"

    def L_(engine, pairs, **kw):
        # packed form: every leg is "a context that scans sample after sample of this kind" -- whatever slot list the leg before built is
        # dropped, and the engine's own rule builds the one this leg's table wants before its second scan (the warm-up step of leg())
        if kw.get("packed", packed) and not args.no_slot_list:
            engine.slot_list(0)
            engine.slot_list(1)
        return leg(engine, wl.out_path, k, e, pairs, kw.pop("n_contigs", nc), cl, packed=kw.pop("packed", packed), **kw)

    try:
        if headline:
            # DEEP FOCUSED: the headline's 100 M pairs from 300 of the 13000 genomes (100x): the table a fifth full, trio-first scan,
            # transfers found and voted -- the same read count where the path has something to find (`value_found`)
            eng.synth_options(0, 20, 300)
            eng.synth_pairs(1, 2, nc, cl, 0, args.pairs, L)
            out["uhgg_deep_focused_sample"] = dict(L_(eng, args.pairs, sample_contigs=300),
                                                   workload="13000x1000000 bp ref, 100 M pairs drawn from 300 of its contigs (100x), sample=1")
            if packed and not args.no_slot_list:       # the same sample with phase B's position-ordered kernel (debug bit 25: the slot list is left unused)
                eng.set_debug(1 << 25)
                d0 = leg(eng, wl.out_path, k, e, args.pairs, nc, cl, packed=True, sample_contigs=300, steps=2, want_stats=False)
                eng.set_debug(0)
                out["uhgg_deep_focused_sample"]["position_ordered_kernel"] = {"value": d0["value"], "ms_per_step": d0["ms_per_step"], "phase_ms": d0["phase_ms"],
                                                                              "scan_B_form": d0["scan_B_form"]["form"],
                                                                              "same_peaks": (d0["raw_peaks"], d0["filtered_peaks"]) == (out["uhgg_deep_focused_sample"]["raw_peaks"], out["uhgg_deep_focused_sample"]["filtered_peaks"])}
            eng.pairs_clear()
            emit("found")
            # READ LENGTH (round 5).  The fast forms of phases A and C take reads of up to 159 bases (128 k-mer offsets);
            # (1) the found-something sample with 1 % of its pairs as 250-base reads: those are passed over by the fast forms and
            # handled by the generic ones -- round 4 sent the whole batch of 16 Mi pairs down the generic paths for one such read;
            eng.synth_read_mix(10, 250)
            eng.synth_pairs(1, 2, nc, cl, 0, args.pairs, L)
            eng.synth_read_mix(0, 0)
            out["uhgg_deep_focused_mixed_1pct_L250"] = dict(L_(eng, args.pairs, sample_contigs=300, steps=2),
                                                            workload="the deep focused sample with 1 % of its pairs as 250-base reads (lhgt_synth_read_mix(10, 250))")
            eng.pairs_clear()
            # (2) the same sample as 250-base reads throughout: the generic scatter of phase A, the generic vote
            eng.synth_pairs(1, 2, nc, cl, 0, args.pairs, 250)
            out["uhgg_deep_focused_L250"] = dict(L_(eng, args.pairs, sample_contigs=300, steps=2, L=250),
                                                 workload="the deep focused sample as 250-base reads (100 M pairs, 166x)")
            eng.pairs_clear()
            emit("read lengths")
            # ... with the reference's own read model: SNPs at 1 % of the sample genomes' bases (species20_snp0.01, test/run_BKP_detection.sh;
            # paper_results/simulation.py:280-299) -- sequencing differences change table fill, the form phase B picks and the vote's survivors
            eng.synth_options(10, 20, 300)
            eng.synth_pairs(1, 2, nc, cl, 0, args.pairs, L)
            out["uhgg_deep_focused_snp1pct"] = dict(L_(eng, args.pairs, sample_contigs=300),
                                                    workload="the same with SNPs at 1 % of the sample genomes' bases (synth_options(10, 20, 300))")
            eng.pairs_clear()
            # the same reference under a 10x sample of those 300 genomes
            fp = 10_000_000
            eng.synth_options(0, 20, 300)
            eng.synth_pairs(1, 2, nc, cl, 0, fp, L)
            out["uhgg_focused_sample"] = dict(L_(eng, fp, sample_contigs=300), workload="13000x1000000 bp ref, 10 M pairs drawn from 300 of its contigs (10x), sample=1")
            eng.pairs_clear()
            # the CLI's default --sample 2000000000 (E:1392-1398): 2e9 / (2 * 100 M * 150) = 6.67 % of the pairs survive the
            # sampling array; any subset of iid pairs is iid, so the kept pairs are generated directly
            kept = int(2e9 / (2 * 150))
            eng.synth_options(0, 20, 0)
            eng.synth_pairs(1, 2, nc, cl, 0, kept, L)
            d = L_(eng, kept)
            d.update(workload=f"configs[2] under the pipeline's default --sample 2000000000: {kept} of 100 M pairs kept (resident; a real run is bound by parsing the other 93 %)",
                     input_pairs=args.pairs, input_pairs_per_s_M=round(args.pairs / (d["ms_per_step"] * 1e-3) / 1e6, 1))
            out["uhgg_default_sample"] = d
            eng.pairs_clear()
            # ... and on a realistic sample: the same 6.67 M pairs drawn from the 300 genomes (a 100 M-pair sample of them under --sample 2000000000)
            eng.synth_options(0, 20, 300)
            eng.synth_pairs(1, 2, nc, cl, 0, kept, L)
            eng.synth_options(0, 20, 0)
            d = L_(eng, kept, sample_contigs=300)
            d.update(workload=f"the deep focused sample under the pipeline's default --sample 2000000000: {kept} of 100 M pairs kept", input_pairs=args.pairs,
                     input_pairs_per_s_M=round(args.pairs / (d["ms_per_step"] * 1e-3) / 1e6, 1))
            out["uhgg_default_sample_focused"] = d
            eng.pairs_clear()
            emit("focused")
            # a RAGGED catalogue: the same 13 Gbase cut into ~118 k contigs (median 4.8 kb, a third shorter than one scan tile) -- what
            # UHGG looks like.  The k - 1 positions without a k-mer at every contig end are contrast peaks (E:931-932, 644-671), so
            # ten times as many peaks register k-mers and phase C is the phase that feels it
            from localhgt_amd.synth import ragged_cuts
            cuts = ragged_cuts(nc * cl)
            eng.synth_reference_cuts(1, nc, cl, cuts)
            eng.synth_pairs(1, 2, nc, cl, 0, args.pairs, L)
            d = L_(eng, args.pairs, n_contigs=len(cuts) - 1, ref_bases=nc * cl, recall=False, steps=2)
            d.update(workload=f"configs[2]'s bases and reads, the reference cut into {len(cuts) - 1} pieces of a catalogue-like length distribution (localhgt_amd.synth.ragged_cuts)")
            out["uhgg_ragged_reference"] = d
            eng.pairs_clear()
            # ... and the ragged catalogue under the deep focused sample: what a real run on a real catalogue looks like
            eng.synth_options(0, 20, 300)
            eng.synth_pairs(1, 2, nc, cl, 0, args.pairs, L)
            d = L_(eng, args.pairs, n_contigs=len(cuts) - 1, ref_bases=nc * cl, recall=False)
            d.update(workload="the ragged catalogue under the deep focused sample: 100 M pairs drawn from the first 300 Mbase of its base stream (100x), sample=1")
            out["uhgg_ragged_deep_focused"] = d
            eng.synth_options(0, 20, 0)
            eng.pairs_clear()
            if packed:
                # INDEX FORM: the headline of rounds 1-4 -- the index file's hashes resident (12 B per base: 156 GB, no room for a slot
                # list), phase B's position-ordered kernels -- and the deep focused sample on it, for the comparison across rounds
                eng.slot_list(0)
                eng.set_reference_form(False)
                eng.synth_reference(1, nc, cl)
                index_bytes = eng.reference_info()["resident_bytes"]
                eng.synth_pairs(1, 2, nc, cl, 0, args.pairs, L)
                d = L_(eng, args.pairs, packed=False, steps=3)
                d.update(workload="configs[2] with the index file's hashes resident (lhgt_set_reference_form(0), --ref-form index): rounds 1-4's headline",
                         resident_index_bytes=index_bytes,
                         same_peaks_as_headline=(d["raw_peaks"], d["filtered_peaks"]) == (detail["raw_peaks"], detail["filtered_peaks"]))
                out["uhgg_index_form"] = d
                eng.pairs_clear()
                eng.synth_options(0, 20, 300)
                eng.synth_pairs(1, 2, nc, cl, 0, args.pairs, L)
                eng.synth_options(0, 20, 0)
                d = L_(eng, args.pairs, packed=False, sample_contigs=300, steps=2)
                d.update(workload="the deep focused sample with the index file's hashes resident: rounds 3-4's value_found")
                out["uhgg_deep_focused_index_form"] = d
                eng.pairs_clear()
                emit("index form")
            else:
                # the headline workload with the reference resident as packed bases (3/8 byte per base instead of 12)
                eng.synth_reference(1, nc, cl)
                index_bytes = eng.reference_info()["resident_bytes"]
                eng.set_reference_form(True)
                eng.synth_reference(1, nc, cl)
                eng.synth_pairs(1, 2, nc, cl, 0, args.pairs, L)
                d = L_(eng, args.pairs, packed=True, steps=2)
                d.update(workload="configs[2] with the reference resident as packed bases (lhgt_set_reference_form(1), LHGT_REF_FORM=packed), hashes recomputed in phase B",
                         resident_reference_bytes=eng.reference_info()["resident_bytes"], resident_index_bytes=index_bytes, slot_list=eng.slot_list(),
                         same_peaks_as_headline=(d["raw_peaks"], d["filtered_peaks"]) == (detail["raw_peaks"], detail["filtered_peaks"]))
                out["uhgg_packed_reference"] = d
                eng.pairs_clear()
    except Exception as ex:
        out["uhgg_error"] = str(ex)[:200]
    eng.close()
    emit("uhgg legs")

    # ---- live PMC of the legs whose roofline is quoted: children of this command, now that this process holds nothing on the GPU
    if headline and not args.no_pmc:
        todo = [("uhgg_deep_focused_sample", dict(sample_contigs=300), {}), ("configs1_1g", dict(workload="1g", contigs=1000, pairs=10_000_000), {})]
        if args.full:
            todo.append(("uhgg_ragged_reference", dict(ragged=True), {}))
        detail["pmc_secondary"] = {}
        for name, over, _ in todo:
            pm, note = collect_pmc(argparse.Namespace(**dict(vars(args), **over)), PMC_PASSES)
            if pm:
                detail["pmc_secondary"][name] = pmc_traffic(pm)
            if note:
                detail["pmc_note"] = "; ".join(x for x in (detail.get("pmc_note"), note) if x)
        for name in ("uhgg_deep_focused_sample", "uhgg_ragged_reference"):
            t = detail["pmc_secondary"].get(name)
            if t and isinstance(out.get(name), dict) and "_shape" in out[name]:
                reroof(out[name], t)
        emit("pmc of the secondary legs")

    try:
        if (args.contigs, args.pairs) != (1000, 10_000_000):
            with Engine(k, e, device=local) as e1:
                e1.rng_seed(1)
                e1.coder_generate()
                e1.synth_reference(1, 1000, 1_000_000)
                e1.synth_pairs(1, 2, 1000, 1_000_000, 0, 10_000_000, L)
                d = leg(e1, wl.out_path, k, e, 10_000_000, 1000, 1_000_000, steps=5, traffic=(detail.get("pmc_secondary") or {}).get("configs1_1g"))
                d["workload"] = "BASELINE configs[1]: 1000x1000000 bp ref, 10 M pairs, k=32 e=3, sample=1"
                out["configs1_1g"] = d
    except Exception as ex:
        out["configs1_error"] = str(ex)[:200]
    try:
        if headline:
            # BASELINE configs[4] names a reference of more than 50 GB, 200 M reads over 8 GPUs, k = 21 / 32.  Its index (12 bytes per
            # base: 600 GB) only fits sharded over the node; packed (3/8 byte per base) the whole 50 Gbase reference, its per-position
            # arrays and the tables fit ONE GPU.  One GPU's share of the reads (25 M pairs) drawn from 300 of the 50 000 genomes.
            nc5, fp = 50_000, 25_000_000
            # ... and (round 6) configs[4] AS NAMED: 200 M input pairs under the CLI's default --sample 2000000000 -- the reference's own answer to
            # a catalogue-sized reference (E:1392-1398: ratio = 2e9 / (2 x 200 M x 150) = 3.33 %): 6 666 666 pairs survive the sampling array
            # whatever the input size, and any subset of iid pairs is iid, so the kept pairs are generated directly
            in5, kept5 = 200_000_000, int(2e9 / (2 * 150))
            legs5, named5 = {}, {}
            for kk in (32, 21):
                with Engine(kk, e, device=local) as e5:
                    e5.rng_seed(1)
                    e5.coder_generate()
                    e5.set_reference_form(True)
                    e5.synth_reference(1, nc5, cl)
                    e5.synth_options(0, 20, 300)
                    e5.synth_pairs(1, 2, nc5, cl, 0, fp, L)
                    d = leg(e5, wl.out_path, kk, e, fp, nc5, cl, steps=2, sample_contigs=300, packed=True)
                    d["resident_reference_bytes"] = e5.reference_info()["resident_bytes"]
                    legs5[f"k{kk}"] = d
                    e5.pairs_clear()
                    e5.synth_pairs(1, 3, nc5, cl, 0, kept5, L)
                    d = leg(e5, wl.out_path, kk, e, kept5, nc5, cl, steps=2, sample_contigs=300, packed=True)
                    d.update(input_pairs=in5, input_pairs_per_s_M=round(in5 / (d["ms_per_step"] * 1e-3) / 1e6, 1))
                    named5[f"k{kk}"] = d
            out["configs4_progenomes_1gpu"] = dict(legs5, workload=f"{nc5}x{cl} bp ref (50 Gbase) resident as packed bases on ONE GPU, 25 M pairs "
                                                                  "(one GPU's share of configs[4]'s 200 M) from 300 of its genomes, e=3, sample=1, k = 32 and 21")
            out["configs4_as_named"] = dict(named5, workload=f"BASELINE configs[4]: {nc5}x{cl} bp ref (50 Gbase, packed, ONE GPU), {in5} input pairs from 300 of its genomes under the "
                                                             f"default --sample 2000000000: {kept5} pairs kept (resident; a real run is bound by parsing the other 96.7 %), e=3, k = 32 and 21")
    except Exception as ex:
        out["configs4_error"] = str(ex)[:200]
    emit("configs[1], configs[4]")
    try:
        if headline and args.full:
            out["pipelined_samples"] = pipelined_samples(k, e, local, nc, cl, args.pairs)
    except Exception as ex:
        out["pipelined_error"] = str(ex)[:200]


def run_e2e(detail, args, local, emit, timeout_s=520):
    """the from-FASTQ legs (tools/benchlib/legs.py: e2e_from_files) in a child of their own, after everything the line must carry:
    they write 23 GB of files and drive the whole host pipeline, and nothing that happens to them may take the measurement along"""
    from . import ROOT
    with tempfile.TemporaryDirectory(prefix="lhgt_e2e_rec_") as tmp:
        rec = os.path.join(tmp, "e2e.json")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--e2e-child", rec, "-k", str(args.k), "-e", str(args.e), "--device", str(local)]
        if args.full:
            cmd.append("--full")
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                 "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        try:
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s, env=env)
            if res.returncode == 0 and os.path.exists(rec):
                detail["e2e"] = json.load(open(rec))
            else:
                detail["e2e"] = dict(json.load(open(rec)) if os.path.exists(rec) else {}, error=f"child rc {res.returncode}", tail=res.stdout.decode(errors="replace")[-600:])
        except subprocess.TimeoutExpired:
            detail["e2e"] = dict(json.load(open(rec)) if os.path.exists(rec) else {}, error=f"child not done after {timeout_s} s")
        except Exception as ex:   # noqa: BLE001
            detail["e2e"] = {"error": str(ex)[:200]}
    emit("e2e")


def e2e_child(path, k, e, device, full):
    """(child of run_e2e): the batch leg first (round 6: the regime the headline is quoted on, reached through a shipped entry point),
    written out at once, then the from-files legs of rounds 2-5"""
    from .legs import e2e_batch
    rec = {}
    try:
        rec["batch_13g"] = e2e_batch(k, e, device)
    except Exception as ex:   # noqa: BLE001
        rec["batch_13g"] = {"error": str(ex)[:300]}
    with open(path, "w") as fh:
        json.dump(rec, fh)
    rec.update(e2e_from_files(k, e, device, full=full))
    with open(path, "w") as fh:
        json.dump(rec, fh)
