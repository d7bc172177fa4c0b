"""CPU: the pair pipeline's core pinning helper never guesses — without a readable GPU topology it returns None (threads stay where the
scheduler puts them), and LCR_PIPE_BIND=0 switches it off."""
import os

from lcrnet_amd import pipeline


def test_no_topology_no_binding(monkeypatch):
    monkeypatch.setattr(pipeline, "gpu_node_cpus", lambda i: None)
    assert pipeline.compact_core_set(0, 8) is None


def test_compact_set_is_a_prefix_of_the_node_and_respects_the_switch(monkeypatch):
    node = list(range(64, 192))
    monkeypatch.setattr(pipeline, "gpu_node_cpus", lambda i: node)
    monkeypatch.delenv("LCR_PIPE_BIND", raising=False)
    assert pipeline.compact_core_set(0, 8) == node[:8]
    monkeypatch.setattr(pipeline, "gpu_node_cpus", lambda i: node[:6])          # a rank that already owns a small share: nothing to compact
    assert pipeline.compact_core_set(0, 8) is None
    monkeypatch.setattr(pipeline, "gpu_node_cpus", lambda i: node)
    monkeypatch.setenv("LCR_PIPE_BIND", "0")
    assert pipeline.compact_core_set(0, 8) is None


def test_gpu_node_cpus_without_a_gpu_is_none():
    assert pipeline.gpu_node_cpus(0) is None or all(c in os.sched_getaffinity(0) for c in pipeline.gpu_node_cpus(0))
