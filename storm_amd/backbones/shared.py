from ..util.registry import Registry

BackboneRegistry = Registry("Backbone")
