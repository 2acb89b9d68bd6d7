"""The driver runs `__graft_entry__.smoke()` on a fresh MI355X before the bench: the GPU suite runs it too, so that a check inside it
that has gone stale (round 5: a bitwise comparison that the f16-plane engine no longer meets) fails HERE first."""
import pytest

pytestmark = pytest.mark.gpu


def test_graft_entry_smoke_passes():
    import __graft_entry__ as entry
    entry.smoke()
