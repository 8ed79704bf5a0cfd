"""bench.py prints ONE JSON line on stdout; the driver reads it from the tail of the run's output.  Round 4's line had
grown to 20 KB and came back unparsed.  Here the line is built from a stubbed measurement with every field at its
longest (all seven configurations, every kernel, long texts) and must stay small and loadable; the full tables go to
the side file."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

KERNELS = ['nam_first', 'nam_step', 'nam_step_sparse', 'gram', 'gram_reduce', 'null_local', 'obs_counts', 'percell_fdr',
           'condition', 'global_test', 'select', 'resid_xb', 'project_xb', 'colsum', 'standardize', 'ncorrs', 'zero_variance',
           'nam_finish', 'batch_kurtosis', 'rowpass16', 'stat_median', 'rccl']


def stub_measurement(name, steps):
    n, N, k, nsteps, Nnull, n_covs = bench.WORKLOADS[name][:6]
    n_batches = bench.WORKLOADS[name][6] if len(bench.WORKLOADS[name]) > 6 else 0
    prof = {kname: (0.123456789 * (i + 1) * steps, steps * (2 if kname == 'gram' else 1)) for i, kname in enumerate(KERNELS)}
    prof['nam_step'] = (8.19215 * steps * n / 2e6, steps)               # the dense walk step dominates, as measured
    return dict(name=name, n=n, N=N, k=k, nsteps=nsteps, Nnull=Nnull, n_covs=n_covs, n_batches=n_batches, nnz=int(40.23 * n), wA=4,
                dt=0.0178123456 * steps, t_cold=0.1084321, i8=(True, 75192, False), t_gen=5.04321, prof=prof, p=0.000999000999000999,
                t_adopt=0.0828765, n_loc=n, nnz_loc=int(40.23 * n), halo=(123456, 234567), sharded_inputs=True, steps=steps, warmup=5,
                comm=('rccl', 8), halo_comm=True, kw={}, dev_bytes=20123456789, pinned=not name.endswith('_unpinned'),
                two_call_path=dict(taken=123456, not_eligible=12345, general=1234, need_pcs=123, stale=12),
                per_rank={k_: [1234.567 + r for r in range(8)] for k_ in ('allreduce_ms', 'halo_exchange_ms', 'halo_wait_ms', 'host_ms',
                                                                           'kernel_ms', 'wall_ms')})


def build_line(world=1):
    args = argparse.Namespace(scaling='strong', partition='populations', comm='rccl', details=None, force_dist=False)
    m = stub_measurement('C4', 20)
    main_sum = bench.summary(m, world, 20)
    extra = {}
    for name in ('C5', 'C3', 'C2', 'C3_default_nsteps', 'C3_covs_batches', 'C4_block8', 'C4_unpinned', 'C4_state_f32', 'C3_state_f32'):
        st = bench.DEFAULT_STEPS[name][0]
        mm = stub_measurement(name, st)
        extra[name] = dict(workload=bench.workload_text(mm, world, args), steps=st, warmup=3, **bench.summary(mm, world, st))
    if world > 1:                                  # the weak-scaling triple of an N > 1 line, with its own per-rank decomposition
        mm = stub_measurement('C4_block8', 50)
        extra['C4_block8_weak'] = dict(workload='w' * 300, steps=50, warmup=10, scaling='weak', cells=250000 * world,
                                       **bench.summary(mm, world, 50))
    extra['C6_failed'] = dict(error=repr(RuntimeError('x' * 1000)))
    stages = dict(nam=43.71, resid_svd=9.31, global_test=0.64, local_test=16.04, percell_apply=143.21)
    cpu = dict(value=9025360.1, unit='cell*perm/s', cores=16, kind='port', mode='reference-cost', seconds=16.62, host_cpus=256,
               blas_threads=16, p_value=0.000999000999000999, stages_s=stages, value_without_percell_apply=25517190.4,
               sample_short='reference-cost port on 150000 cells x 200 samples, k=30, nsteps=3, Nnull=1000, same generator seed 0',
               sample='y' * 700, extrapolated_to_workload=dict(cells=2000000, seconds=212.8, stages_s=stages, value=9400341.4))
    cpu_c2 = dict(cpu, gpu_ms_per_step=1.568, gpu_over_cpu=16000.1, same_p_value_as_gpu=True)
    cpu_c2.pop('extrapolated_to_workload')
    details = bench.assemble_details(m, main_sum, cpu, cpu_c2, extra, world, 20, 5, args)
    return bench.contract_line(details), details


def test_contract_line_is_small_and_loads():
    line, details = build_line()
    assert '\n' not in line
    assert len(line) < 8000, len(line)
    assert len(line) < 6144, len(line)           # the target; the driver's tail is 8 KB
    assert len(json.dumps(details)) > len(line)  # the tables are in the side file, not lost
    d = json.loads(line)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['steps'] == 20 and d['warmup'] == 5 and d['n_gpus'] == 1
    assert d['vs_baseline'] is None and d['dtype'] == 'f64' and d['data'] == 'synthetic'
    assert set(d['config']) >= {'workload', 'parallelism', 'arithmetic'} and 'model' not in d['config']
    assert len(d['config']['workload']) <= 200 and len(d['config']['arithmetic']) <= 120
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in d['roofline'], key
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in d['cpu_baseline'], key
    assert d['cpu_baseline_C2_full']['seconds'] and 'extrapolated' not in d['cpu_baseline_C2_full']
    assert set(d['other_configs']) == {'C5', 'C3', 'C2', 'C3_default_nsteps', 'C3_covs_batches', 'C4_block8', 'C4_unpinned',
                                       'C4_state_f32', 'C3_state_f32', 'C6_failed'}
    assert 'ranks' not in d                        # (one GPU: nothing to decompose across ranks)
    for name, o in d['other_configs'].items():
        if name != 'C6_failed':
            assert o['ms_per_step'] > 0 and o['value'] > 0 and 'frac' in o['roofline']


def test_roofline_fraction_of_the_line_is_priced_on_the_survey_bytes():
    """a diffusion step t >= 2 is nnz (4 + w_A) + 8 (n + 1) + 8 n + 16 n N bytes; the line's `frac` is that over
    the launch time, whatever else the launch writes (the fused selection by-product goes to `frac_fused`)."""
    line, details = build_line()
    d = json.loads(line)
    assert d['roofline']['kernel'] == 'nam_step'
    m = stub_measurement('C4', 20)
    want = m['nnz'] * 8 + 8 * (m['n'] + 1) + 8 * m['n'] + 16 * m['n'] * m['N']
    avg_s = d['roofline']['avg_us'] * 1e-6
    assert abs(d['roofline']['frac'] - want / avg_s / 1e9 / 8000.0) < 2e-4
    assert abs(d['roofline']['achieved'] - d['roofline']['frac'] * 8000.0) < 1.0
    assert d['roofline']['frac_fused'] >= d['roofline']['frac']


def test_line_with_ranks():
    line, _ = build_line(world=8)
    d = json.loads(line)
    assert len(line) < 6144
    assert d['n_gpus'] == 8 and d['config']['comm'] == 'rccl' and d['config']['comm_ranks'] == 8
    # what a rank's step is made of, [max over ranks, rank 0], so that a scaling curve explains itself
    for key in ('kernel_ms', 'halo_wait_ms', 'halo_exchange_ms', 'allreduce_ms', 'host_ms', 'wall_ms'):
        assert len(d['ranks'][key]) == 2 and d['ranks'][key][0] >= d['ranks'][key][1] > 0, key
    assert d['ranks']['halo_rows_out_in_rank0'] == [123456, 234567] and len(d['ranks']['halo_mb_per_exchange_out_in_rank0']) == 2
    assert 'rccl_transport' in d
    w = d['other_configs']['C4_block8_weak']
    assert w['scaling'] == 'weak' and w['cells'] == 2000000 and w['ranks']['kernel_ms'][0] > 0 and w['ms_per_step'] > 0


def test_transport_lines_of_an_rccl_log_are_counted(tmp_path, monkeypatch):
    log = tmp_path / 'nccl.log'
    log.write_text('host:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC\n'
                   'host:1:1 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC\n'
                   'host:1:1 [0] NCCL INFO Channel 00/0 : 1[1] -> 0[0] [receive] via NET/Socket/0\n'
                   'host:1:1 [0] NCCL INFO Channel 00 : 0[0] -> 1[1] via SHM/direct/direct\n'
                   'host:1:1 [0] NCCL INFO comm 0x1 rank 0 nranks 2 - Init COMPLETE\n')
    monkeypatch.setenv('CNA_BENCH_NCCL_LOG', str(log))
    assert bench.rccl_transport(0) == {'P2P/IPC': 2, 'NET': 1, 'SHM/direct': 1}
