"""BASELINE.json configs[4]: steady-state stream + injected leader failures -> re-elections.

Per tick, after the dense ack tick, every group fails with probability p (Bernoulli from
the counter-based hash, salt 7): the local (leader) instance crashes and restarts
(JG_CMD_RESTART = Raft::new + Chain::new on the persisted tree: State::default(), head =
id_gen = commit), times out with voted_for == None (the only way a node can campaign in the
reference), and receives granted VoteResponses from the next R/2 replicas
(what their follower.rs:97-101 can_vote answers for a node with the highest commit) — all as
explicit command rows through jg_submit/jg_step.  What happens next is the reference's truth:
the re-elected leader's first append trips assert!(id > head) (chain.rs:163, Q8) whenever it
had committed anything, so failed groups drop out until they are restarted again.
"""
import numpy as np

from josefine_amd import capi

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_hash(seed, tick, gg, r):
    """hash(seed, tick, g, r) of DESIGN.md "Synthetic traces" (vectorised over gg)."""
    with np.errstate(over="ignore"):
        a = mix64(np.uint64(seed) + np.uint64(tick) * np.uint64(0x9E3779B97F4A7C15))
        return mix64(a ^ (np.asarray(gg, dtype=np.uint64) * np.uint64(8) + np.uint64(r)))


def failure_rows(seed, tick, group_base, G, R, node_ids, self_slots, percent=1):
    """Command columns of one tick's failures (kwargs for submit_columns)."""
    gg = np.arange(G, dtype=np.uint64) + np.uint64(group_base)
    failing = np.nonzero(synth_hash(seed, tick, gg, 7) % np.uint64(100) < np.uint64(percent))[0].astype(np.uint32)
    n = len(failing)
    votes = R // 2
    ids = np.array(node_ids, dtype=np.uint32)
    kind = [np.full(n, capi.CMD_RESTART, np.uint8), np.full(n, capi.CMD_TIMEOUT, np.uint8)]
    group = [failing, failing]
    frm = [np.zeros(n, np.uint32), np.zeros(n, np.uint32)]
    for k in range(1, votes + 1):
        kind.append(np.full(n, capi.CMD_VOTE_RESPONSE, np.uint8))
        group.append(failing)
        frm.append(ids[(self_slots[failing].astype(np.int64) + k) % R])
    kind, group, frm = np.concatenate(kind), np.concatenate(group), np.concatenate(frm)
    return dict(kind=kind, group=group, from_=frm, term=np.ones(len(kind), np.uint64),
                flag=np.ones(len(kind), np.uint8)), n
