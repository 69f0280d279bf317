"""Drop-in for the `simple_knn` package the reference imports at src/model/gaussian_model.py:5."""
