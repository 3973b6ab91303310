"""Turns two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate runs: the TCC block cannot hold both) into the
per-launch HBM traffic record profiles/rNN_pmc_traffic.json that bench.py reports as roofline.traffic.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/x_counter_collection.csv gpurun_out/pmc_write/x_counter_collection.csv profiles/rNN_pmc_traffic.json [bench.json]

The optional fourth argument is the bench JSON line of the SAME command: its config.workload is stored as `_workload`, and
bench.py quotes roofline.traffic from the file only on that workload.

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are KiB; FETCH_SIZE
tallies 128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE is exact."""
import csv, json, sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name']
        for key in ('lstm_fwd_p3', 'lstm_bwd_p3', 'lstm_fwd_p2', 'lstm_bwd_p2', 'dec_fwd_persist', 'dec_bwd_persist', 'dec_fwd_stream', 'dec_bwd_stream', 'att_bwd_energy_kernel',
                    'att_energy_kernel', 'gemm16_nt_kernel', 'gemm16_tn_kernel', 'gemm_kernel'):
            if key in name:
                a = acc.setdefault(key, [0, 0.0])
                a[0] += 1
                a[1] += float(r['Counter_Value'])
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    workload = None
    if len(sys.argv) > 4:
        lines = [l for l in open(sys.argv[4]).read().splitlines() if l.startswith('{')]
        workload = json.loads(lines[-1])['config']['workload']
    f, w = per_kernel(fetch, 'FETCH_SIZE'), per_kernel(write, 'WRITE_SIZE')
    rec = {}
    for k in sorted(set(f) | set(w)):
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        fa = sf / nf if nf else 0.0
        wa = sw / nw if nw else 0.0
        rec[k] = {'launches': max(nf, nw), 'FETCH_SIZE_KiB_avg': fa, 'WRITE_SIZE_KiB_avg': wa,
                  'hbm_bytes_per_launch': (2.0 * fa + wa) * 1024.0, 'hbm_bytes_per_launch_uncorrected': (fa + wa) * 1024.0}
    if workload is not None:
        rec['_workload'] = workload
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
