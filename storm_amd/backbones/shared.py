from ..util.registry import Registry

BackboneRegistry = Registry("Backbone")
for _name in ("convtasnet", "gagnet", "ae-ncsnpp"):
    BackboneRegistry.declare_out_of_scope(_name, "only the NCSN++ family is on the reverse-SDE sampling path this engine covers (BASELINE.json north_star)")
