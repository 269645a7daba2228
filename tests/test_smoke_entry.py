"""The driver's round-end entry point must keep working: run it as a GPU test."""
import pytest


@pytest.mark.gpu
def test_graft_entry_smoke(cuda):
    import __graft_entry__
    __graft_entry__.smoke()
