"""GRAFX graph object (mirrors grafx.data.graph.GRAFX — reference src/grafx/data/graph.py:12-302).

Kept a ``networkx.MultiDiGraph`` subclass with the same ``graph``-dict
attributes so user code that builds graphs against the reference runs
unchanged.
"""
import warnings

import networkx as nx
import torch


class GRAFX(nx.MultiDiGraph):
    def __init__(self, config=None, invalid_op="error"):
        if invalid_op not in ("error", "warn", "mute"):
            raise Exception(f"Incorrect invalid_op is given: {invalid_op}.")
        super().__init__()
        self.graph = dict(
            counter=0,
            consecutive_ids=True,
            batch=False,
            config=config,
            config_hash=hash(config),
            invalid_op=invalid_op,
            rendering_order_method=None,
            type_sequence=None,
        )

    # -- construction -----------------------------------------------------------------
    def add(self, node_type, parameters=None, name=None):
        """Add one node; returns its id (reference graph.py:101-129)."""
        cfg = self.graph["config"]
        if cfg is not None and node_type not in cfg.node_types:
            self.raise_warning(f"Invalid node_type: {node_type}, this graph only allows {cfg.node_types}.")
            return None
        node_id = self.graph["counter"]
        assert node_id not in self.nodes()
        self.add_node(node_id, node_type=node_type, parameters=parameters, name=name)
        self.graph["counter"] = node_id + 1
        return node_id

    def remove(self, node_id):
        """Remove a node; returns (incoming, outgoing) edges (reference graph.py:131-146)."""
        incoming = list(self.in_edges(node_id, data=True))
        outgoing = list(self.out_edges(node_id, data=True))
        self.remove_node(node_id)
        self.graph["consecutive_ids"] = False
        return incoming, outgoing

    def connect(self, source_id, dest_id, outlet="main", inlet="main"):
        """Add an edge after the same validity checks as the reference (graph.py:148-195)."""
        if self.has_edge(source_id, dest_id):
            for existing in self.get_edge_data(source_id, dest_id).values():
                if existing["outlet"] == outlet and existing["inlet"] == inlet:
                    self.raise_warning(f"{source_id} <{outlet}> -> {dest_id} <{inlet}>: existing edge.")
        if source_id == dest_id:
            self.raise_warning("no self edge is allowed!")
        cfg = self.graph["config"]
        if cfg is not None:
            src_type = self.nodes[source_id]["node_type"]
            outlets = cfg.node_type_dict[src_type]["outlets"]
            if outlet not in outlets:
                self.raise_warning(f"Provided outlet: '{outlet}', while {src_type} only accepts {outlets}.")
                return
            dst_type = self.nodes[dest_id]["node_type"]
            inlets = cfg.node_type_dict[dst_type]["inlets"]
            if inlet not in inlets:
                self.raise_warning(f"Provided inlet: '{inlet}', while {dst_type} only accepts {inlets}.")
                return
        self.add_edge(source_id, dest_id, outlet=outlet, inlet=inlet)

    def add_serial_chain(self, node_list):
        """Add nodes connected head-to-tail; returns (first_id, last_id) (graph.py:197-222).

        Dict entries are forwarded to :meth:`add` as keyword arguments (the
        reference drops the returned id for dict entries, graph.py:213-214 —
        here the id is kept so the chain is connected as documented).
        """
        first = prev = None
        for entry in node_list:
            node_id = self.add(entry) if isinstance(entry, str) else self.add(**entry)
            if prev is not None:
                self.connect(prev, node_id)
            if first is None:
                first = node_id
            prev = node_id
        return first, prev

    def raise_warning(self, message):
        mode = self.graph["invalid_op"]
        if mode == "error":
            raise Exception(message)
        if mode == "warn":
            warnings.warn("Following operation is invalid: " + message)
        elif mode != "mute":
            raise AssertionError(mode)

    # -- printable form ---------------------------------------------------------------
    def __str__(self):
        def port(tag, right=False):
            if tag == "main":
                return ""
            return f"<{tag}> " if right else f" <{tag}>"

        lines = [f"GRAFX with {self.number_of_nodes()} nodes & {self.number_of_edges()} edges"]
        for i, data in self.nodes(data=True):
            head = f"  [{i}] {data['node_type']}"
            outs = list(self.out_edges([i], data=True))
            if len(outs) == 1:
                _, to, e = outs[0]
                head += f"{port(e['outlet'])} -> {port(e['inlet'], True)}[{to}] {self.nodes[to]['node_type']}"
                lines.append(head)
            else:
                lines.append(head)
                for _, to, e in outs:
                    tag = f"<{e['outlet']}>" if e["outlet"] != "main" else ""
                    lines.append(f"    {tag} -> {port(e['inlet'], True)}[{to}] {self.nodes[to]['node_type']}")
        return "\n".join(lines)

    # -- attribute views on self.graph (reference graph.py:235-302) ---------------------
    @property
    def counter(self):
        return self.graph["counter"]

    @counter.setter
    def counter(self, val):
        assert isinstance(val, int)
        self.graph["counter"] = val

    @property
    def consecutive_ids(self):
        return self.graph["consecutive_ids"]

    @consecutive_ids.setter
    def consecutive_ids(self, val):
        assert isinstance(val, bool)
        self.graph["consecutive_ids"] = val

    @property
    def batch(self):
        return self.graph["batch"]

    @batch.setter
    def batch(self, val):
        assert isinstance(val, bool)
        self.graph["batch"] = val

    @property
    def config(self):
        return self.graph["config"]

    @config.setter
    def config(self, val):
        raise Exception("config can be setted after the initialization.")

    @property
    def config_hash(self):
        return self.graph["config_hash"]

    @config_hash.setter
    def config_hash(self, val):
        raise Exception("config_hash cannot be setted directly.")

    @property
    def invalid_op(self):
        return self.graph["invalid_op"]

    @invalid_op.setter
    def invalid_op(self, val):
        assert isinstance(val, str)
        self.graph["invalid_op"] = val

    @property
    def rendering_order_method(self):
        return self.graph["rendering_order_method"]

    @rendering_order_method.setter
    def rendering_order_method(self, val):
        assert isinstance(val, str)
        self.graph["rendering_order_method"] = val

    @property
    def type_sequence(self):
        return self.graph["type_sequence"]

    @type_sequence.setter
    def type_sequence(self, val):
        assert isinstance(val, (list, torch.LongTensor))
        self.graph["type_sequence"] = val
