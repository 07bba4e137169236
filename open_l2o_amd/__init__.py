"""open_l2o_amd -- MI355X-native inner unroll loop of Open-L2O's coordinate-wise
LSTM optimizers (L2O-DM / L2O-RNNProp), behind the reference's own
MetaOptimizer / networks / problems / util API.  See DESIGN.md.
"""
__version__ = "0.1.0"
