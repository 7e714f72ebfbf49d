gridToVTK = None
pointsToVTK = None
