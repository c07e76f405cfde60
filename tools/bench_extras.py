"""Side figures of bench.py's JSON line that are not part of its contract (the contract -- metric, value, roofline, cpu_baseline -- stays in bench.py):
  probe_cut_proxy   one GPU playing shard 0 of S: the plain target-sharded pass against the probe-list cut
  files_to_files    the module boundary: query / target DBs on disk -> alignment DB on disk, fresh processes / SP_SERVER=auto / a resident server (tools/module_e2e.py),
                    at the bench proxy's density and at 2e8 residues, next to the reference's module wall time estimated from bench.py's cpu_baseline samples
Called from bench.py at N = 1 on the default workload only; nothing here touches oracle/."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def probe_cut_proxy(sa, args, shards=8, spacers=100000, probes_per_query=0):
    """one GPU as shard 0 of `shards` (configs[3]: 100 000 spacers against 1e7 residues cut into target shards): HIP-event time of the matching kernel when the shard
    enumerates every query itself (sp_ctx_prefilter_candidates) against the probe-list cut -- enumerate 1 / shards of the queries against the union of all shards'
    occupancy maps (sp_ctx_probe_enumerate), place the probes the `shards` enumerating ranks address to shard 0 and fetch (sp_ctx_probe_fetch); the candidates of both
    ways are compared record for record.  All shard contexts live on this GPU (their maps make the union; their enumerations produce what shard 0 receives)."""
    import ctypes as C  # noqa: F401
    import bench
    from spacepharer_amd.dist import decompose_by_residues
    tres, toff, qres, qoff = bench.make_workload(spacers, args.targets, args.target_len, 12345)
    lens = np.diff(toff).astype(np.int64) + 2
    nq = len(qoff) - 1
    ctxs = []
    for r in range(shards):
        start, count = decompose_by_residues(lens, r, shards)
        c = sa.SearchContext()
        c.set_targets_shard_raw(tres.tobytes(), bench.u64p(toff), len(toff) - 1, start, start + count)
        c.set_queries_raw(qres.tobytes(), bench.u64p(qoff), nq)
        c.set_timing(2)
        c.set_tuning("match_blocks_per_cu", -1)      # (all that fit, as for the ONE context a rank of the sharded job has: the library leaves room for other contexts' aligners when it sees several searching on a device)
        if probes_per_query:
            c.set_tuning("probe_per_query", probes_per_query)      # (room per destination: a dense shard holds a third of all k-mer slots)
        ctxs.append(c)
    c0 = ctxs[0]
    plain_ms = []
    for _ in range(6):
        c0.prefilter_candidates()
        plain_ms.append(c0.stat("ms_match"))
    plain = np.ascontiguousarray(c0.candidates())
    maps = [c.shard_occupancy() for c in ctxs]
    for c in ctxs:
        c.probe_set_shards(maps)
    qlens = np.diff(qoff).astype(np.int64) + 2
    recs, per_query, enum_ms = [], np.zeros(nq, dtype=np.uint32), []
    for r, c in enumerate(ctxs):
        a, n = decompose_by_residues(qlens, r, shards)
        for _ in range(4 if r == 0 else 1):
            out = c.probe_enumerate(a, a + n)
            if r == 0:
                enum_ms.append(c.stat("ms_match"))
        recs.append(out[0][0])
        per_query[a:a + n] = out[0][1]
    R = np.concatenate(recs)
    fetch_ms = []
    for _ in range(6):
        c0.probe_fetch(R, per_query)
        fetch_ms.append(c0.stat("ms_match"))
    cut = np.ascontiguousarray(c0.candidates())
    equal = sorted(bytes(x) for x in cut) == sorted(bytes(x) for x in plain)
    for c in ctxs:
        c.close()
    real = int((R[:, 0] != 0xFFFFFFFF).sum())
    return {"what": "one GPU as shard 0 of %d, %d spacers (%d peptides) against 1 / %d of %.1e residues: the matching kernel when the shard enumerates every query itself against "
                    "enumerate 1 / %d of the queries (against the union of the shards' maps) + fetch of the probes the %d enumerating ranks address to this shard; HIP events, the "
                    "other shard contexts idle on the same GPU; the exchange the cut adds has not run on xGMI" % (shards, spacers, nq, shards, float(toff[-1]), shards, shards),
            "plain_k_match_grp_ms": round(min(plain_ms[2:]), 4), "cut_enumerate_ms": round(min(enum_ms[1:]), 4), "cut_fetch_ms": round(min(fetch_ms[2:]), 4),
            "cut_matching_ms": round(min(enum_ms[1:]) + min(fetch_ms[2:]), 4), "x_plain": round(min(plain_ms[2:]) / (min(enum_ms[1:]) + min(fetch_ms[2:])), 3),
            "probes_received_by_the_shard": real, "probe_bytes_received": real * 16, "candidates": int(len(plain)), "candidates_equal_plain_pass": bool(equal)}


def files_to_files(out, args, dense, cores, nq):
    """adds out["module_e2e"], out["files_to_files"] (and its "dense" block when the dense line was measured)"""
    try:
        pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "module_e2e.py"), "10000", "100000"], capture_output=True, text=True, timeout=1200)
        doc = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])["module_e2e"]
        out["module_e2e"] = {"what": "spacer-queries/s files-to-files on the bench targets (1e7 residues): module argv in, alignment DB on disk out; 'process' = fresh processes of "
                                     "spacepharer_amd/spacepharer-amd per module call (process start + HIP runtime start-up + tables included), 'warm' = repeated call inside one process, 'served' = client processes of the same binary forwarding to a resident `spacepharer-amd serve` (runtime, tables and target set stay up), 'auto_server' = the same fresh client processes with SP_SERVER=auto and nothing set up: the first one starts the server (first = the pair of calls that pays for it, next = the pairs after it; the server leaves when idle)",
                             "runs": [{"spacers": r["spacers"], "mode": r["mode"], "threads": r["threads"], "process_s": r["process_total_s"], "process_spacer_queries_per_s": r["process_spacer_queries_per_s"],
                                       "in_process_s": [x["total_s"] for x in r["runs"]], "warm_spacer_queries_per_s": r["warm_spacer_queries_per_s"],
                                       "served_s": r["served_total_s"], "served_spacer_queries_per_s": r["served_spacer_queries_per_s"],
                                       "auto_server_s": r["auto_server_total_s"], "auto_server_first_spacer_queries_per_s": r["auto_server_first_spacer_queries_per_s"],
                                       "auto_server_next_spacer_queries_per_s": r["auto_server_next_spacer_queries_per_s"], "aln_db_bytes": r["aln_db_bytes"],
                                       "aln_db_crc": r["aln_db_crc"]} for r in doc]}
        cb = out.get("cpu_baseline", {})
        if cb.get("kind") == "reference" and cb.get("module_value"):
            # the reference at the same boundary, scaled from the sample: index build + tables once per prefilter process, search loops per spacer
            for run in out["module_e2e"]["runs"]:
                ref_s = cb["index_build_s"] + cb["kmer_tables_s"] + cb["search_loops_s"] * (run["spacers"] * 6.0 / max(1, nq if args.cpu_sample >= 3000 else min(args.cpu_sample * cores, nq)))
                run["reference_module_wall_s_estimate"] = ref_s
                run["x_reference"] = {"process": ref_s / min(run["process_s"]), "auto_server_first": ref_s / run["auto_server_s"][0], "auto_server_next": ref_s / min(run["auto_server_s"][1:]),
                                      "served": ref_s / min(run["served_s"][1:])}
        # SURVEY section 8d's own boundary -- "DBs resident -> result DBs written" -- as named values: 100 000 spacers, module calls on DB triples
        f2f = {"what": "files to files at 100 000 spacers (query / target DBs on disk -> alignment DB on disk; module argv as the workflow scripts pass it); the bench's `value` is the "
                       "search step with inputs resident in HBM and results delivered into page-locked host memory", "spacers": 100000}
        for run in out["module_e2e"]["runs"]:
            if run["spacers"] != 100000:
                continue
            tag = "fused_search" if "search" in run["mode"] or "fused" in run["mode"] else "prefilter_then_align"
            f2f[tag] = {"fresh_processes_spacer_queries_per_s": run["process_spacer_queries_per_s"], "served_spacer_queries_per_s": run["served_spacer_queries_per_s"],
                        "auto_server_next_spacer_queries_per_s": run["auto_server_next_spacer_queries_per_s"], "x_reference": run.get("x_reference"),
                        "reference_module_wall_s_estimate": run.get("reference_module_wall_s_estimate")}
        out["files_to_files"] = f2f
    except Exception as e:      # noqa: BLE001
        out["module_e2e"] = {"error": repr(e)[:300]}
    if dense is not None and "files_to_files" in out:
        # the target statement's own configuration: 100 000 spacers against the density of GenBank_phage_2018_09 (2e8 residues), files to files, next to what the
        # reference's module pair pays there (index build + similar-k-mer tables once per prefilter process + the search loops, from the dense reference sample)
        try:
            pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "module_e2e.py"), "100000", "--targets", "1000000", "--light"], capture_output=True, text=True, timeout=1500)
            doc = json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])["module_e2e"]
            dcb = out["dense"].get("cpu_baseline", {})
            ref_s = None
            if dcb.get("kind") == "reference" and dcb.get("search_loops_s"):
                n_d = min(len(dense["batches"][0][3]) - 1, 200 * cores) if args.cpu_sample > 0 else None
                ref_s = dcb["index_build_s"] + dcb["kmer_tables_s"] + dcb["search_loops_s"] * (600000.0 / n_d) if n_d else None
            fd = {"what": "files to files, 100 000 spacers (6e5 peptides) vs 1 000 000 ORFs x 200 aa = 2e8 residues (query / target DBs on disk -> alignment DB on disk); "
                          "fresh = one fresh process of the front end per module call, auto_server = the same calls with SP_SERVER=auto (first pays for the server's start), served = a resident server",
                  "spacers": 100000, "target_residues": 200000000, "reference_module_wall_s_estimate": ref_s,
                  "reference_estimate_from": "dense.cpu_baseline: index build + similar-k-mer tables + search loops scaled from the sample to 6e5 peptides (DB I/O and text not counted)"}
            for run in doc:
                tag = "fused_search" if run["mode"] == "fused" else "prefilter_then_align"
                ent = {"fresh_process_s": run["process_total_s"], "fresh_processes_spacer_queries_per_s": run["process_spacer_queries_per_s"],
                       "auto_server_s": run["auto_server_total_s"], "auto_server_next_spacer_queries_per_s": run["auto_server_next_spacer_queries_per_s"],
                       "served_s": run["served_total_s"], "served_spacer_queries_per_s": run["served_spacer_queries_per_s"], "in_process_s": [x["total_s"] for x in run["runs"]],
                       "aln_db_bytes": run["aln_db_bytes"], "aln_db_crc": run["aln_db_crc"]}
                if ref_s:
                    ent["x_reference"] = {"fresh_process": ref_s / min(run["process_total_s"]), "auto_server_first": ref_s / run["auto_server_total_s"][0],
                                          "auto_server_next": ref_s / min(run["auto_server_total_s"][1:]), "served": ref_s / min(run["served_total_s"][1:])}
                if run.get("fresh_process_laps"):
                    ent["fresh_process_laps"] = run["fresh_process_laps"]
                fd[tag] = ent
            out["files_to_files"]["dense"] = fd
        except Exception as e:      # noqa: BLE001
            out["files_to_files"]["dense"] = {"error": repr(e)[:300]}
