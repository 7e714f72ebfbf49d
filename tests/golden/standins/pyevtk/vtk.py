VtkGroup = None
