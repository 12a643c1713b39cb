"""Drop-in operator surface with the reference's module and callable names.

`from keras_smpl.batch_smpl import SMPLLayer` etc. (model.py:6-11 of the reference) map to
`<this package>.keras_smpl.batch_smpl.SMPLLayer`; see INTEGRATION.md.
"""
