"""Node-type configuration (mirrors grafx.data.configs — reference src/grafx/data/configs.py:3-126)."""

_IN = {"inlets": [], "outlets": ["main"]}
_OUT = {"inlets": ["main"], "outlets": []}
_SISO = {"inlets": ["main"], "outlets": ["main"]}
UTILITY_TYPES = ["in", "out", "mix"]
UTILITY_DICT = {"in": _IN, "out": _OUT, "mix": _SISO}


class NodeConfigs:
    """Registry of node types with their inlets/outlets; utility types come first.

    ``config`` is a list of type names (all SISO) or a dict
    ``{type: {"inlets": [...], "outlets": [...]}}`` (reference configs.py:33-42).
    """

    def __init__(self, config):
        if isinstance(config, list):
            table = {}
            for name in UTILITY_TYPES + config:
                table[name] = UTILITY_DICT.get(name, _SISO)
        elif isinstance(config, dict):
            table = {**UTILITY_DICT, **config}
        else:
            raise ValueError("Invalid type for config.")
        self._index(table)

    def _index(self, table):
        self.node_type_dict = table
        self.node_types = list(table)
        self.num_node_types = len(table)
        self.node_type_to_index = {t: i for i, t in enumerate(self.node_types)}
        self.num_inlets = {t: len(c["inlets"]) for t, c in table.items()}
        self.num_outlets = {t: len(c["outlets"]) for t, c in table.items()}
        widest_in = max([1] + list(self.num_inlets.values()))
        widest_out = max([1] + list(self.num_outlets.values()))
        self.siso_only = widest_in == 1 and widest_out == 1
        if not self.siso_only:
            self.max_num_inlets, self.max_num_outlets = widest_in, widest_out
            self.inlet_to_index = {t: {n: i for i, n in enumerate(c["inlets"])} for t, c in table.items()}
            self.outlet_to_index = {t: {n: i for i, n in enumerate(c["outlets"])} for t, c in table.items()}

    # public helpers of the reference class (configs.py:71-120), for code that re-registers types on an instance
    def get_default_config(self, node_type):
        """Port layout a bare type name stands for: sources have no inlet, sinks no outlet, everything else is SISO."""
        return UTILITY_DICT.get(node_type, _SISO) if node_type in ("in", "out") else _SISO

    def unpack_list(self, node_type_list):
        """(Re)build the registry from type names alone (every type gets its default port layout)."""
        self._index({name: self.get_default_config(name) for name in node_type_list})

    def unpack_dict(self, node_type_dict):
        """(Re)build the registry from ``{type: {"inlets": [...], "outlets": [...]}}``."""
        self._index(dict(node_type_dict))

    def __getitem__(self, node_type):
        return self.node_type_dict[node_type]

    def __str__(self):
        def fmt(ports):
            return "None" if not ports else "<" + ", ".join(ports) + ">"

        lines = [f"NodeConfigs with {self.num_node_types} node types (siso_only={self.siso_only})"]
        for t, c in self.node_type_dict.items():
            lines.append(f"  ({self.node_type_to_index[t]}) {t}: {fmt(c['inlets'])} -> {fmt(c['outlets'])}")
        return "\n".join(lines)
