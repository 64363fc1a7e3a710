"""Drop-in `simple_knn` package: `from simple_knn._C import distCUDA2`
(street_gaussian/models/gaussian_model.py:5, gaussian_model_actor.py:10,
data_processor/utils/render_utils.py:6) resolves to the HIP implementation."""
