"""jax.sharding stand-ins for big_vision/sharding.py (oracle/run_reference_sharding.py): a PartitionSpec that is a LEAF for
the tree functions (not a tuple subclass), NamedSharding as a (mesh, spec) record, Mesh as a name -> size mapping."""


class PartitionSpec:
  def __init__(self, *axes):
    self.axes = tuple(axes)

  def __iter__(self):
    return iter(self.axes)

  def __len__(self):
    return len(self.axes)

  def __eq__(self, other):
    return tuple(self) == tuple(other)

  def __repr__(self):
    return f"PartitionSpec{self.axes}"


class NamedSharding:
  def __init__(self, mesh, spec):
    self.mesh, self.spec = mesh, spec


class Mesh:
  """Only what the sharding rules read: `mesh.shape[axis_name]`."""

  def __init__(self, shape):
    self.shape = dict(shape)
