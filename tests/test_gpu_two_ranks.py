"""The multi-GPU classes (yag_slam_amd/dist.py) with two REAL matchers -- two rank processes, both on cuda:0, over gloo with the
records staged through the host (tests/two_rank_scenarios.py; the processes are forked by tests/conftest.py before the session
touches the GPU).  What the 2-rank gloo tests of tests/test_dist_gloo.py run against stand-in matchers and the 1-rank RCCL tests
cannot show: chain-id bases, uneven and empty shards, a tie across ranks, the post-expansion record and the in-place all-gather of
the angle split, on device pointers, with world = 2.  (Reference: the serial chain loop of /root/reference/yag_slam/graph_slam.py:217-254.)"""
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_loop_matcher_two_real_matchers(two_ranks):
    r0, r1 = two_ranks("sharded_loop")
    assert r0 == r1                      # both ranks agree on the winner records ...
    assert r0["all"][1] == float(r0["expected_winner"])  # ... the tie across the ranks went to the lowest global chain id
    assert r0["one"][1] == 0.0


def test_sharded_loop_matcher_post_expansion_record(two_ranks):
    r0, r1 = two_ranks("sharded_expansion")
    assert r0 == r1


@pytest.mark.parametrize("stress", [False, True])
def test_angle_split_two_real_matchers(two_ranks, stress):
    r0, r1 = two_ranks("angle_split", stress=stress)
    assert r0 == r1
