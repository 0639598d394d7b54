"""hostbudget: per-process shares of a node-level host budget (reference: one process, 32 workers, main_1v.py:124)."""
import os

from pointnetgpd_amd import hostbudget as hb


def test_workers_are_a_node_total(monkeypatch):
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert hb.local_world() == 8
    assert hb.workers_per_rank(32) == 4 and hb.workers_per_rank(0) == 0 and hb.workers_per_rank(3) == 1
    assert 8 * hb.workers_per_rank(32) == 32                     # what the reference's one process started
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert hb.local_world() == 2 and hb.workers_per_rank(32) == 16
    monkeypatch.delenv("WORLD_SIZE")
    assert hb.local_world() == 1 and hb.workers_per_rank(32) == 32


def test_threads_follow_affinity_quota_and_ranks(monkeypatch):
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    monkeypatch.setattr(hb, "_cgroup_quota", lambda: None)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert hb.cpus() == 64 and hb.threads_per_rank() == 8
    monkeypatch.setattr(hb, "_cgroup_quota", lambda: 16)          # a container limited to 16 CPUs of the 64 it can see
    assert hb.cpus() == 16 and hb.threads_per_rank() == 2
    monkeypatch.setattr(hb, "_cgroup_quota", lambda: 4)
    assert hb.threads_per_rank() == 1                             # never zero
    assert 8 * hb.threads_per_rank() <= 8
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    monkeypatch.setattr(hb, "_cgroup_quota", lambda: None)
    assert hb.threads_per_rank() == 8 and hb.threads_per_rank(cap=4) == 4


def test_eig_pool_uses_the_rank_share(monkeypatch):
    from pointnetgpd_amd import gpg
    monkeypatch.setattr(hb, "cpus", lambda: 16)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert gpg._eig_workers() == 2


def test_cli_divides_num_workers(monkeypatch):
    from pointnetgpd_amd import mains
    args = mains.build_parser().parse_args(["--mode", "train", "--batch-size", "64", "--synthetic", "128",
                                            "--num-workers", "32"])
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    mains._make_loaders(mains.VARIANTS["1v"], args, world=1)
    assert args.rank_workers == 32                                # single process: the flag as given
    args.num_workers = 0                                          # (world > 1 needs an initialised process group for the
    mains._make_loaders(mains.VARIANTS["1v"], args, world=1)      #  sampler; the split itself is hostbudget's, tested above)
    assert args.rank_workers == 0
