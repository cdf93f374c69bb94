"""Summarise rocprofv3 --pmc result databases (rocpd sqlite) into mean counter values per dispatch and kernel.
Usage: python tools/summarize_pmc_db.py out.json run1/pmc_results.db [run2/pmc_results.db ...]  (prints the summary too)"""
import collections, json, re, sqlite3, sys


def short(name):
    m = re.search(r"(ntt_strided<[12], (?:true|false)|ntt_contig|ntt_mid|tile_dense|tile_loop|tile_step|pair_accumulate|plan_tiles|propose_lattice|field_update|propose|apply|claim|field_sites|copy16|cells_to_slots|derive_slots|tile_parts)", name)
    return m.group(1) if m else name.split("(")[0][:40]


def tables(con):
    names = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    return {re.sub(r"_[0-9a-f]{8}_[0-9a-f_]+$", "", n): n for n in names}


def summarise(paths):
    out = collections.defaultdict(dict)
    for path in paths:
        con = sqlite3.connect(path)
        T = tables(con)
        q = (f"select s.kernel_name, p.name, e.value, d.id from {T['rocpd_pmc_event']} e "
             f"join {T['rocpd_info_pmc']} p on e.pmc_id = p.id "
             f"join {T['rocpd_kernel_dispatch']} d on e.event_id = d.event_id "
             f"join {T['rocpd_info_kernel_symbol']} s on d.kernel_id = s.id order by d.id")
        per = collections.OrderedDict()               # (kernel, counter, dispatch) -> sum over the counter's instances (XCDs / SEs)
        for kname, cname, value, did in con.execute(q):
            key = (short(kname), cname, did)
            per[key] = per.get(key, 0.0) + float(value)
        acc = collections.defaultdict(list)
        for (k, c, _), v in per.items():
            acc[(k, c)].append(v)
        for (k, c), v in acc.items():
            v = v[len(v) // 5:]                       # drop warm-up dispatches
            out[k][c + "_KB" if c in ("FETCH_SIZE", "WRITE_SIZE") else c] = sum(v) / len(v)
            out[k]["dispatches"] = len(v)
    return out


if __name__ == "__main__":
    res = summarise(sys.argv[2:])
    json.dump({"note": "rocprofv3 --pmc (counters in their own passes, --kernel-trace only), mean per dispatch after dropping the "
                       "first fifth; FETCH_SIZE / WRITE_SIZE in KB as reported (gfx950: FETCH_SIZE counts 64 B per 128-B request "
                       "of wide streaming reads, see MI355X_MICROARCH.md; these kernels read 4-16 B per lane: uncalibrated)",
               "per_dispatch": res}, open(sys.argv[1], "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))
