"""Disjoint union of graphs (mirrors grafx.data.batch.batch_grafx — reference src/grafx/data/batch.py:4-37)."""
import networkx as nx


def batch_grafx(G_list):
    offset, running, shifted = 0, [], []
    first_hash = None
    for i, G in enumerate(G_list):
        if not G.consecutive_ids:
            raise Exception("The node ids must be consecutive.")
        if G.batch:
            raise Exception(f"Graph of index {i} is already a batched graph.")
        if first_hash is None:
            first_hash = G.config_hash
        elif first_hash != G.config_hash:
            raise Exception("Graphs with different node configs cannot be batched.")
        if i:
            G = nx.relabel_nodes(G, {n: n + offset for n in range(G.number_of_nodes())})
        shifted.append(G)
        offset += G.counter
        running.append(offset)
    G_batch = nx.union_all(shifted)
    G_batch.graph["counter"] = running  # list of per-graph cumulative counters, as upstream
    G_batch.graph["batch"] = True
    return G_batch
